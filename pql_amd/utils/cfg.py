"""Hydra-optional config loader over pql_amd/cfg/*.yaml.

The reference composes its config with Hydra 1.x (`@hydra.main(config_path=pql/cfg, config_name="default")`,
scripts/train_pql.py:27).  Hydra/omegaconf are not installed in this image, so this module implements the
subset the reference uses: `defaults` lists (group: option, `file.yaml` includes, `_self_`), relative
interpolation `${.key}`, and command-line overrides `a.b=value` / `+a.b=value` / `++a.b=value` / `group=option` with Hydra's
struct-mode rule (overriding a key the composed config lacks is an error, not a new key).  The result is a `Cfg`
with attribute and item access, accepted everywhere the reference passes a DictConfig.
"""
from __future__ import annotations

from pathlib import Path

import yaml

CFG_DIR = Path(__file__).resolve().parent.parent / "cfg"


class Cfg(dict):
    """dict with attribute access (stand-in for omegaconf.DictConfig)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        import copy
        return Cfg({k: copy.deepcopy(v, memo) for k, v in self.items()})


_SCI = __import__("re").compile(r"^[+-]?(\d+\.?\d*|\.\d+)[eE][+-]?\d+$")


def _wrap(x):
    if isinstance(x, str) and _SCI.match(x):   # PyYAML reads "5e6" as a string; Hydra/omegaconf as a float
        return float(x)
    if isinstance(x, dict):
        return Cfg({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def _load_file(path: Path, group_choices=None):
    """Load one yaml, resolving its `defaults` list relative to its own directory."""
    data = yaml.safe_load(path.read_text()) or {}
    defaults = data.pop("defaults", [])
    out = {}
    self_done = False
    for d in defaults:
        if d == "_self_":
            _merge(out, data)
            self_done = True
        elif isinstance(d, str):                       # plain include: "actor_critic.yaml"
            _merge(out, _load_file(path.parent / d))
        elif isinstance(d, dict):                      # group: option
            (group, option), = d.items()
            if group_choices and group in group_choices:
                option = group_choices[group]
            if option is None:
                out.setdefault(group, None)
                continue
            fpath = (path.parent / group / f"{option}.yaml" if not str(option).endswith(".yaml")
                     else path.parent / group / option)
            if group == "task" and not fpath.exists():
                # the reference's `task=<IsaacGymEnvs task name>` (pql/cfg/default.yaml:7-9 + the isaacgymenvs search
                # path): here every task is the synthetic vectorised env with that task's shapes
                from pql_amd.envs.synthetic import TASK_SHAPES   # (torch-only module; imported here to keep cfg import light)
                if str(option) not in TASK_SHAPES:   # a typo must fail at config load, as Hydra's missing-config error does
                    raise ValueError(f"task={option}: no such task; known tasks: {', '.join(sorted(TASK_SHAPES))} "
                                     f"(or task=synthetic with task.obs_dim / task.act_dim)")
                sub = _load_file(path.parent / group / "synthetic.yaml")
                sub["name"] = str(option)
            else:
                sub = _load_file(fpath)
            out[group] = _merge(out.get(group) or {}, sub)
    if not self_done:
        _merge(out, data)
    return out


def _parse_value(text):
    try:
        return yaml.safe_load(text)
    except yaml.YAMLError:
        return text


def _resolve(node, root):
    """`${.key}` -> sibling key; `${a.b}` -> absolute path."""
    for k, v in list(node.items()):
        if isinstance(v, dict):
            _resolve(v, root)
        elif isinstance(v, str) and v.startswith("${") and v.endswith("}"):
            ref = v[2:-1]
            if ref.startswith("."):
                node[k] = node[ref[1:]]
            else:
                cur = root
                for part in ref.split("."):
                    cur = cur[part]
                node[k] = cur


def load_cfg(overrides=(), config_name="default", cfg_dir: Path = CFG_DIR) -> Cfg:
    """Compose the config.  overrides: iterable of "key=value" strings (Hydra command-line syntax)."""
    groups = {p.name for p in cfg_dir.iterdir() if p.is_dir()}
    group_choices, assigns = {}, []
    for ov in overrides:
        key, _, val = ov.partition("=")
        plus = len(key) - len(key.lstrip("+"))   # Hydra: `key=` overrides an existing key, `+key=` adds one, `++key=` does either
        key = key.lstrip("+")
        if key in groups:
            group_choices[key] = _parse_value(val)
        else:
            assigns.append((key, _parse_value(val), plus))
    tree = _load_file(cfg_dir / f"{config_name}.yaml", group_choices)
    for key, val, plus in assigns:
        cur = tree
        parts = key.split(".")
        exists = True
        for p in parts[:-1]:
            if not isinstance(cur.get(p), dict):
                exists = False
                cur[p] = {}
            cur = cur[p]
        exists = exists and parts[-1] in cur
        # a typo in an override must fail at config load, as it does under Hydra's struct mode -- not train a default silently
        if plus == 0 and not exists:
            raise KeyError(f"Could not override '{key}': no such key in the composed config.  To add a new key use +{key}={val}")
        if plus == 1 and exists:
            raise KeyError(f"Could not append '{key}': the config already has it.  To override it drop the '+', or use ++{key}={val}")
        cur[parts[-1]] = val
    _resolve(tree, tree)
    return _wrap(tree)
