"""Checkpoint helpers in the reference's on-disk format (pql/utils/model_util.py:24-36, evaluator.py:112-119):
`torch.save({'obs_rms': (mean, var, eps) | None, 'actor': state_dict, 'critic': state_dict}, path)` with the
reference's state_dict key names, so files are interchangeable.  The reference's W&B artifact upload / download is
out of scope (no network): `load_model` reads a local file only."""
from __future__ import annotations

import torch


def save_model(path, actor, critic, rms, wandb_run=None, description=None):
    """actor / critic: modules or state_dicts; rms: RunningMeanStd.get_states() tuple or None."""
    sd = lambda m: m.state_dict() if hasattr(m, "state_dict") else m  # noqa: E731
    cpu = lambda d: {k: v.detach().cpu() for k, v in d.items()}       # noqa: E731
    ckpt = {"obs_rms": None if rms is None else tuple(t.detach().cpu() if torch.is_tensor(t) else t for t in rms),
            "actor": cpu(sd(actor)), "critic": cpu(sd(critic))}
    torch.save(ckpt, path)
    return ckpt


def load_model(model, model_type, path):
    """Load `model_type` in {'actor', 'critic', 'obs_rms'} from a local checkpoint into `model`
    (a pql_amd module, or a RunningMeanStd for 'obs_rms').  Tensors only: weights_only=True."""
    weights = torch.load(path, map_location="cpu", weights_only=True)
    if model_type not in weights:
        raise KeyError(f"invalid model type: {model_type}")
    if model_type == "obs_rms":
        if weights[model_type] is None:
            return False
        mean, var, eps = weights[model_type]
        model.mean, model.var = mean.to(model.mean.device), var.to(model.var.device)
        model.epsilon = float(eps)
        return True
    model.load_state_dict(weights[model_type])
    return True
