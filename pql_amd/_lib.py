"""ctypes binding of libpqlk.so (include/pqlk.h).

The library is the product: if it is missing or fails to load this module raises, there is no
CPU or eager-PyTorch fallback anywhere in pql_amd.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import torch

_HERE = Path(__file__).resolve().parent
LIB_FILE = Path(os.environ.get("PQLK_LIB", _HERE / "csrc" / "libpqlk.so"))   # PQLK_LIB: A/B a tuning build

MAX_LAYERS = 8
ACT_NONE, ACT_TANH, ACT_TANH_NOISE = 0, 1, 2


class PqlReplayDesc(C.Structure):
    _fields_ = [("records", C.c_void_p), ("capacity", C.c_int64), ("obs_dim", C.c_int32), ("act_dim", C.c_int32),
                ("rec_ld", C.c_int32), ("reserved", C.c_int32)]


class PqlMlpDesc(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("n_nets", C.c_int32), ("dims", C.c_int32 * (MAX_LAYERS + 1))]


_P, _I64, _I32, _F = C.c_void_p, C.c_int64, C.c_int32, C.c_float

# name -> (restype, argtypes); kept in the order of include/pqlk.h
PROTOTYPES = {
    "pqlk_version": (C.c_int, []),
    "pqlk_strerror": (C.c_char_p, [C.c_int]),
    "pqlk_ld": (_I64, [_I64]),
    "pqlk_replay_rec_ld": (_I64, [_I32, _I32]),
    "pqlk_replay_insert": (C.c_int, [C.POINTER(PqlReplayDesc), _I64, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P]),
    "pqlk_replay_gather": (C.c_int, [C.POINTER(PqlReplayDesc), _P, _I64, _P, _P, _P, _P, _P, _P]),
    "pqlk_replay_gather_fused": (C.c_int, [C.POINTER(PqlReplayDesc), _P, _I64, _P, _P, _F, C.c_int, _P, _I64, _P, _P, _I64, _P, _P, _P]),
    "pqlk_philox_draws": (C.c_int, [C.c_uint64, _I64, _I32, _P, _I64, _P, _I64, _P, _I64, _I32, _I32, _P]),
    "pqlk_philox_increment": (_I64, [_I64]),
    "pqlk_nstep_push_emit": (C.c_int, [_P, _I64, _I32, _I32, _I32, _I64, _I64, _P, _P, _P, _P, _P, C.POINTER(C.c_float),
                                       _P, _P, _P, _P, _P, C.POINTER(C.c_int64), _P]),
    "pqlk_mlp_param_floats": (_I64, [C.POINTER(PqlMlpDesc)]),
    "pqlk_mlp_net_stride": (_I64, [C.POINTER(PqlMlpDesc)]),
    "pqlk_mlp_layer_offsets": (C.c_int, [C.POINTER(PqlMlpDesc), _I32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pqlk_mlp_acts_floats": (_I64, [C.POINTER(PqlMlpDesc), _I64]),
    "pqlk_mlp_act_offset": (C.c_int, [C.POINTER(PqlMlpDesc), _I64, _I32, _I32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pqlk_mlp_bwd_ws_floats": (_I64, [C.POINTER(PqlMlpDesc), _I64, _I32]),
    "pqlk_mlp_packed_floats": (_I64, [C.POINTER(PqlMlpDesc)]),
    "pqlk_mlp_pack": (C.c_int, [C.POINTER(PqlMlpDesc), _P, _P, _P]),
    "pqlk_mlp_forward": (C.c_int, [C.POINTER(PqlMlpDesc), _P, _P, _I32, _P, _I64, _I64, _I32, _P, _F, _F, _P, _P, _I64, _P]),
    "pqlk_mlp_backward": (C.c_int, [C.POINTER(PqlMlpDesc), _P, _P, _I64, _I64, _P, _P, _P, _I32, _P, _I64, _I32, _I32, _P, _I64,
                                    _P, _I64, _P]),
    "pqlk_mlp_backward_norm": (C.c_int, [C.POINTER(PqlMlpDesc), _P, _P, _I64, _I64, _P, _P, _P, _I32, _P, _I64, _I32, _I32, _P, _I64,
                                         _P, _I64, _P, _P, _P]),
    "pqlk_mlp_norm_parts": (_I32, [C.POINTER(PqlMlpDesc)]),
    "pqlk_mlp_backward_td": (C.c_int, [C.POINTER(PqlMlpDesc), _P, _P, _I64, _I64, _P, _P, _P, _P, _F, _P, _P, _I32, _P, _I64, _P, _P, _P]),
    "pqlk_td_head_loss_parts": (_I32, [C.POINTER(PqlMlpDesc), _I64]),
    "pqlk_td_forward_loss_parts": (_I32, [C.POINTER(PqlMlpDesc), _I64]),
    "pqlk_mlp_forward_td": (C.c_int, [C.POINTER(PqlMlpDesc), _P, _P, _P, _I64, _I64, _P, _P, _P, _P, _F, _P, _P, _I64, _I32, _P]),
    "pqlk_mlp_backward_td_tail": (C.c_int, [C.POINTER(PqlMlpDesc), _P, _P, _I64, _I64, _P, _P, _I32, _P, _I64, _P, _P, _P]),
    "pqlk_mlp_backward_layers": (C.c_int, [C.POINTER(PqlMlpDesc), _P, _P, _I64, _I64, _P, _P, _P, _P, _P, _F, _P, _P, _I32, _P, _I64, _I32, _I32, _P]),
    "pqlk_dpg_backward_ws_floats": (_I64, [C.POINTER(PqlMlpDesc), _I64]),
    "pqlk_dpg_critic_backward": (C.c_int, [C.POINTER(PqlMlpDesc), _P, _P, _I64, _I64, _P, _P, _P, _I64, _I32, _I32, _P, _I64, _P, _P, _I64, _P]),
    "pqlk_dpg_fused_ok": (_I32, [C.POINTER(PqlMlpDesc), C.POINTER(PqlMlpDesc), _I64]),
    "pqlk_dpg_fused_loss_parts": (_I32, []),
    "pqlk_dpg_fused_head_parts": (_I32, [_I64]),
    "pqlk_dpg_fused_mn_offset": (_I64, [C.POINTER(PqlMlpDesc), _I64]),
    "pqlk_mlp_forward_qc": (C.c_int, [C.POINTER(PqlMlpDesc), _P, _P, _I32, _P, _I64, _P, _I64, _I32, _I64, _P, _P, _P]),
    "pqlk_dpg_backward_fused": (C.c_int, [C.POINTER(PqlMlpDesc), _P, _P, _I64, _I64, _P, _P, _P, _I64, _I32, _P, _I64, _P, _P, _I64,
                                          C.POINTER(PqlMlpDesc), _P, _P, _P, _I64, _I32, _P]),
    "pqlk_mlp_backward_tail": (C.c_int, [C.POINTER(PqlMlpDesc), _P, _P, _I64, _I64, _P, _P, _I32, _P, _I64, _P, _P, _I32, _P, _P]),
    "pqlk_dpg_loss_owner": (C.c_int, [_P, _I64, _I32, _P, _I64, _P, _P, _P, _I32, _P, _P, _P]),
    "pqlk_td_mse_loss": (C.c_int, [_P, _P, _I64, _P, _P, _F, _I64, _P, _P, _P, _I32, _P, _P]),
    "pqlk_c51_bce_loss": (C.c_int, [_P, _P, _I64, _I32, _P, _P, _P, _F, _F, _F, _I64, _P, _P, _P, _I32, _P, _P, _P]),
    "pqlk_c51_project": (C.c_int, [_P, _P, _P, _P, _F, _F, _F, _I32, _I64, _P, _P]),
    "pqlk_dpg_loss": (C.c_int, [_P, _I64, _I32, _P, _I64, _P, _P, _P, _I32, _P, _P]),
    "pqlk_clip_adamw_polyak": (C.c_int, [_P, _P, _P, _P, _P, _I64, _F, _F, _F, _F, _F, _F, _F, _F, _P, _P, _P, _P]),
    "pqlk_clip_adamw_polyak_pack": (C.c_int, [C.POINTER(PqlMlpDesc), _P, _P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _F, _F, _F, _F, _P, _P, _P, _P]),
    "pqlk_adamw_polyak_fused": (C.c_int, [C.POINTER(PqlMlpDesc), _P, _P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _F, _F, _F, _F, _P, _P, _P,
                                          _I32, _P, _I32, _F, _P, _I32, _P]),
    "pqlk_loss_parts": (_I32, [_I64, _I32]),
    "pqlk_polyak": (C.c_int, [_P, _P, _I64, _F, _P]),
    "pqlk_sg_head_forward": (C.c_int, [_P, _I64, _P, _I64, _I32, _P, _I64, _P, _P]),
    "pqlk_sg_head_backward": (C.c_int, [_P, _I64, _P, _P, _I64, _P, _I64, _P, _F, _I64, _I32, _P, _P]),
    "pqlk_sac_entropy_shift": (C.c_int, [_P, _I64, _I64, _I32, _P, _P, _I64, _P]),
    "pqlk_sac_alpha_terms": (C.c_int, [_P, _I64, _P, _F, _P, _P, _P, _P, _I32, _P]),
    "pqlk_bn_elu_forward": (C.c_int, [_P, _I64, _I64, _I32, _P, _P, _P, _P, _F, _I32, _F, _P, _P, _P, _P]),
    "pqlk_bn_elu_backward": (C.c_int, [_P, _P, _P, _I64, _I64, _I32, _P, _P, _P, _F, _P, _P, _P, _P, _P]),
    "pqlk_synth_env_step": (C.c_int, [_I64, _I32, _I32, C.c_uint32, C.c_uint32, C.c_uint32, _F, _P, _P, _P, _P, _P]),
    "pqlk_rollout_step": (C.c_int, [_I64, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _P]),
    "pqlk_batch_moments": (C.c_int, [_P, _I64, _I64, _I32, _P, _P, _P, _P]),
    "pqlk_rms_merge": (C.c_int, [_P, _P, _P, _P, _F, _F, _F, _I32, _P, _P, _P]),
    "pqlk_rms_normalize": (C.c_int, [_P, _I64, _I32, _P, _P, _F, _P, _I64, _P]),
    "pqlk_action_noise": (C.c_int, [_P, _P, _P, _F, _I64, _I32, _F, _F, _P, _P]),
}


def _load():
    if not LIB_FILE.exists():
        raise ImportError(
            f"{LIB_FILE} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C {LIB_FILE.parent}`. pql_amd has no fallback path.")
    lib = C.CDLL(os.fspath(LIB_FILE))
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()


class PqlkError(RuntimeError):
    pass


def check(rc: int):
    if rc != 0:
        raise PqlkError(f"{lib.pqlk_strerror(rc).decode()} (rc={rc})")


def ld(cols: int) -> int:
    return int(lib.pqlk_ld(int(cols)))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream(device=None):
    """hipStream_t of torch's current stream: kernels run in order with torch ops and can be graph-captured."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu(t: torch.Tensor, name="tensor"):
    if not t.is_cuda:
        raise PqlkError(f"{name} must live on an MI355X device (got {t.device}); pql_amd has no CPU path")
    if t.dtype not in (torch.float32, torch.int64, torch.int32):
        raise PqlkError(f"{name}: unsupported dtype {t.dtype}")
    if not t.is_contiguous():
        raise PqlkError(f"{name} must be contiguous")
    return t


def mlp_desc(dims, n_nets=1) -> PqlMlpDesc:
    dims = [int(x) for x in dims]
    if not (2 <= len(dims) <= MAX_LAYERS + 1):
        raise PqlkError("MLP needs between 1 and 8 Linear layers")
    d = PqlMlpDesc()
    d.n_layers, d.n_nets = len(dims) - 1, int(n_nets)
    for i, x in enumerate(dims):
        d.dims[i] = x
    return d
