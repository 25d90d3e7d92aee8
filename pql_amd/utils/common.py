"""Host-side helpers with the names the reference's entry points import from `pql.utils.common`
(pql/utils/common.py): class-name plugin registry (:34-42, :79-100), Tracker (:103-126), learner-side
normalize (:139-145), preprocess_cfg / check_device (:148-182, :279-287), handle_timeout (:195-202),
set_random_seed (:45-54), capture_keyboard_interrupt (:185-192).  No wandb / loguru / omegaconf / gym
dependency: logging is stdout + JSONL (see pql_amd.utils.logger)."""
from __future__ import annotations

import ast
import importlib.util
import random
import signal
import sys
from collections import deque
from pathlib import Path

import numpy as np
import torch


# ---------------------------------------------------------------------------- plugin registry
def list_class_names(dir_path):
    """{class name: file} for every top-level class in the *.py files under dir_path (AST scan, no import)."""
    table = {}
    for py in sorted(Path(dir_path).rglob("*.py")):
        if py.name == "__init__.py" or not py.is_file():
            continue
        tree = ast.parse(py.read_text(encoding="utf-8"))
        for node in tree.body:
            if isinstance(node, ast.ClassDef):
                table[node.name] = py
    return table


def load_class_from_path(cls_name, path):
    """Execute the file at `path` as module MOD<cls_name> and return its attribute `cls_name`.
    Like the reference, the module is also registered as sys.modules[cls_name] so pickles resolve."""
    spec = importlib.util.spec_from_file_location(f"MOD{cls_name}", path)
    module = importlib.util.module_from_spec(spec)
    sys.modules[cls_name] = module
    spec.loader.exec_module(module)
    return getattr(module, cls_name)


# ---------------------------------------------------------------------------- bookkeeping
class Tracker:
    """Moving window of the last `max_len` values, pre-filled with zeros."""

    def __init__(self, max_len):
        self.max_len = max_len
        self.moving_average = deque([0] * max_len, maxlen=max_len)

    def __repr__(self):
        return repr(self.moving_average)

    def update(self, value):
        if isinstance(value, (np.ndarray, torch.Tensor)):
            self.moving_average.extend(value.tolist())
        elif isinstance(value, (list, tuple)):
            self.moving_average.extend(value)
        else:
            self.moving_average.append(value)

    def mean(self):
        return np.mean(self.moving_average)

    def std(self):
        return np.std(self.moving_average)

    def max(self):
        return np.max(self.moving_average)


def set_random_seed(seed=None):
    if seed is None:
        seed = random.randint(0, 2 ** 32 - 1)
    np.random.seed(seed)
    torch.manual_seed(seed)
    random.seed(seed)
    return seed


def capture_keyboard_interrupt():
    def _bye(signum, frame):
        print("You pressed Ctrl+C!")
        sys.exit(0)

    signal.signal(signal.SIGINT, _bye)


def handle_timeout(dones, info):
    """done &= ~truncated when the env reports time-limit truncation."""
    trunc = info.get("TimeLimit.truncated", None) if isinstance(info, dict) else None
    if trunc is not None:
        dones = dones * (~trunc)
    return dones


def get_action_dim(action_space):
    if hasattr(action_space, "n"):
        return action_space.n
    return action_space.shape[0]


# ---------------------------------------------------------------------------- math
def normalize(input, normalize_tuple):
    """Learner-side observation normalisation with +-5 clamp; identity when the tuple is None.
    (In the learners this is fused into the replay gather kernel; this stand-alone form serves callers
    that hold a plain tensor.)"""
    if normalize_tuple is None:
        return input
    mean, var, eps = normalize_tuple
    return torch.clamp((input - mean.float()) / torch.sqrt(var.float() + eps), min=-5.0, max=5.0)


# ---------------------------------------------------------------------------- cfg mutation
TASK_REWARD_SCALE = dict(AllegroHand=0.01, Ant=0.01, Humanoid=0.01, Anymal=1., FrankaCubeStack=0.1, ShadowHand=0.01,
                         BallBalance=0.1)
TASK_MAX_TIME = dict(AllegroHand=4800, Ant=3600, Humanoid=3600, Anymal=1800, FrankaCubeStack=3600, ShadowHand=4800,
                     BallBalance=3600)


def check_device(cfg):
    """Sim is always GPU 0; learner GPUs must exist.  (The reference's asserts here are vacuous strings.)"""
    wanted = {0, int(cfg.algo.p_learner_gpu), int(cfg.algo.v_learner_gpu)}
    bad = [g for g in wanted if g >= max(int(cfg.available_gpus), 1)]
    if bad:
        raise ValueError(f"Invalid GPU id(s) {bad}: only {cfg.available_gpus} device(s) visible")


def preprocess_cfg(cfg):
    cfg.available_gpus = torch.cuda.device_count()
    if cfg.algo.name == "PQL":
        check_device(cfg)
    task_name = cfg.task.name if getattr(cfg, "task", None) is not None else None
    if task_name in TASK_REWARD_SCALE and cfg.algo.reward_scale == 1:
        cfg.algo.reward_scale = TASK_REWARD_SCALE[task_name]
    if task_name in TASK_MAX_TIME and cfg.max_time == 3600:
        cfg.max_time = TASK_MAX_TIME[task_name]


class ClassIndex(dict):
    """{class name: defining file} for a package directory -- the reference's string-keyed plugin table
    (`model_name_to_path` / `alg_name_to_path`), built by an AST scan so nothing is imported until selected."""

    def __init__(self, package_file):
        super().__init__(list_class_names(Path(package_file).resolve().parent))

    def resolve(self, name):
        """Load and return the class registered under `name`."""
        return load_class_from_path(name, self[name])
