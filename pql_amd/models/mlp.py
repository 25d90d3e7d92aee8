"""MLP actor / critic family on fused fp32-MFMA kernels.

Drop-in for the hot classes of the reference's `pql/models/mlp.py`: `MLPNet` (:27-40),
`TanhMLPPolicy` (:177-179), `DoubleQ` (:186-203), `DistributionalDoubleQ` (:244-267) -- same
constructor arguments, same methods (`forward`, `get_q1_q2`, `get_q_min`, `get_q1`), same
`state_dict()` key names (`net.{0,2,4,6}.{weight,bias}`, `net_q{1,2}.net.*`) so reference checkpoints
load unchanged.  Selected by class name through `pql_amd.models.model_name_to_path`, exactly like the
reference's plugin lookup (pql/models/__init__.py:5-6).

MI355X design: each module owns ONE flat fp32 parameter arena (layout in include/pqlk.h: per net, per
layer W (out, ld(in)) then b (ld(out)), 128-byte padded) instead of 8-16 nn.Parameters.  Twin critics
are two nets in one arena evaluated by one grouped launch per layer.  Forward/backward go through
pqlk_mlp_forward / pqlk_mlp_backward (autograd-integrated via FusedMlpFn); the learners bypass autograd
and drive the same entry points directly on the arenas.
"""
from __future__ import annotations

import contextlib
import ctypes as C
from collections import OrderedDict
from collections.abc import Sequence

import torch
import torch.nn as nn

from pql_amd import _lib as L

HIDDEN_DEFAULT = (512, 256, 128)  # reference default, mlp.py:32-33


def _first(dim):
    return int(dim[0]) if isinstance(dim, Sequence) else int(dim)


class ArenaLayout:
    """Offsets of every (net, layer) W / b block inside the flat arena (mirrors pqlk_mlp_layer_offsets)."""

    def __init__(self, dims, n_nets):
        self.dims = [int(d) for d in dims]
        self.n_nets = int(n_nets)
        self.desc = L.mlp_desc(self.dims, self.n_nets)
        self.n_layers = len(self.dims) - 1
        self.net_stride = int(L.lib.pqlk_mlp_net_stride(C.byref(self.desc)))
        self.total = int(L.lib.pqlk_mlp_param_floats(C.byref(self.desc)))
        self.w_off, self.b_off = [], []
        for l in range(self.n_layers):
            w, b = C.c_int64(), C.c_int64()
            L.check(L.lib.pqlk_mlp_layer_offsets(C.byref(self.desc), l, C.byref(w), C.byref(b)))
            self.w_off.append(w.value)
            self.b_off.append(b.value)
        self.ld_in = L.ld(self.dims[0])
        self.ld_out = L.ld(self.dims[-1])

    def weight(self, arena, net, l):
        out_f, in_f = self.dims[l + 1], self.dims[l]
        o = net * self.net_stride + self.w_off[l]
        return arena[o: o + out_f * L.ld(in_f)].view(out_f, L.ld(in_f))[:, :in_f]

    def bias(self, arena, net, l):
        o = net * self.net_stride + self.b_off[l]
        return arena[o: o + self.dims[l + 1]]

    def acts_floats(self, B):
        return int(L.lib.pqlk_mlp_acts_floats(C.byref(self.desc), B))

    def act_offset(self, B, net, layer):
        off, ld = C.c_int64(), C.c_int64()
        L.check(L.lib.pqlk_mlp_act_offset(C.byref(self.desc), B, net, layer, C.byref(off), C.byref(ld)))
        return off.value, ld.value

    def bwd_ws_floats(self, B, splits):
        return int(L.lib.pqlk_mlp_bwd_ws_floats(C.byref(self.desc), B, splits))

    def count_params(self):
        return self.n_nets * sum(self.dims[l + 1] * self.dims[l] + self.dims[l + 1] for l in range(self.n_layers))


def default_splits(B: int, cap: int = 16) -> int:
    """Batch splits of the dW GEMM: enough (net x tile x split) blocks to cover 256 CUs, >= 512 rows each (cap: `algo.dw_splits`)."""
    return int(max(1, min(int(cap), B // 512)))


def pad_cols(x: torch.Tensor, ld: int) -> torch.Tensor:
    """(B, c) -> contiguous (B, ld) with zero pad columns (GEMM operands need 128-byte rows)."""
    if x.shape[1] == ld and x.is_contiguous():
        return x
    out = torch.zeros((x.shape[0], ld), dtype=torch.float32, device=x.device)
    out[:, : x.shape[1]] = x
    return out


class PackedWeights:
    """Fragment-ordered copy of an arena's hidden-layer weights for the fused forward kernel (pqlk_mlp_pack).
    `tensor` is None when the layout cannot take the fused path (then forward falls back to per-layer GEMMs).
    refresh() must run after every change of the arena (optimiser step, Polyak, weight hand-off)."""

    def __init__(self, layout: ArenaLayout, device):
        n = int(L.lib.pqlk_mlp_packed_floats(C.byref(layout.desc)))
        self.layout = layout
        self.tensor = torch.zeros(n, dtype=torch.float32, device=device) if n > 0 else None

    def refresh(self, arena):
        if self.tensor is not None:
            L.check(L.lib.pqlk_mlp_pack(C.byref(self.layout.desc), L.ptr(arena), L.ptr(self.tensor), L.stream(arena.device)))
        return self


def mlp_forward_raw(layout: ArenaLayout, arena, x_pad, out_act=L.ACT_NONE, draw=None, noise_std=0.0, noise_clip=0.0,
                    acts=None, out2=None, packed: PackedWeights = None, stash_all=True):
    """Launch the forward; returns the activation stash (last block = (n_nets, B, ld_out) output).
    packed: PackedWeights refreshed from `arena` -> fused hidden layers; None -> one GEMM launch per layer.
    stash_all=2 (PQLK_STASH_OUTPUT_ONLY): `acts` is the (n_nets, B, ld_out) output block itself, nothing else is written."""
    B = x_pad.shape[0]
    dev = x_pad.device
    if acts is None:
        acts = torch.empty(layout.acts_floats(B), dtype=torch.float32, device=dev)
    pk = packed.tensor if packed is not None else None
    with torch.cuda.device(dev):
        L.check(L.lib.pqlk_mlp_forward(C.byref(layout.desc), L.ptr(arena), L.ptr(pk), 2 if stash_all == 2 else (1 if stash_all else 0), L.ptr(x_pad),
                                       x_pad.stride(0), B, out_act, L.ptr(draw), float(noise_std), float(noise_clip),
                                       L.ptr(acts), L.ptr(out2), out2.stride(0) if out2 is not None else 0, L.stream(dev)))
    return acts


def output_view(layout: ArenaLayout, acts, B):
    off, ld = layout.act_offset(B, 0, layout.n_layers - 1)
    return acts[off: off + layout.n_nets * B * ld].view(layout.n_nets, B, ld)


class FusedMlpFn(torch.autograd.Function):
    """autograd bridge: y = MLP(x) for all nets of an arena; backward through pqlk_mlp_backward."""

    @staticmethod
    def forward(ctx, x, arena, layout, out_act):
        L.require_gpu(arena, "parameter arena")
        x_pad = pad_cols(x.to(torch.float32), layout.ld_in)
        acts = mlp_forward_raw(layout, arena, x_pad, out_act)
        B = x.shape[0]
        y = output_view(layout, acts, B)[:, :, : layout.dims[-1]]
        ctx.layout, ctx.out_act, ctx.B, ctx.in_cols = layout, out_act, B, x.shape[1]
        ctx.save_for_backward(x_pad, arena, acts)
        return y.clone()

    @staticmethod
    def backward(ctx, gy):
        x_pad, arena, acts = ctx.saved_tensors
        lay, B = ctx.layout, ctx.B
        dev = x_pad.device
        y = output_view(lay, acts, B)
        dy = torch.zeros((lay.n_nets, B, lay.ld_out), dtype=torch.float32, device=dev)
        g = gy.to(torch.float32)
        if ctx.out_act == L.ACT_TANH:  # dL/dpre = dL/dy * (1 - y^2)
            yv = y[:, :, : lay.dims[-1]]
            g = g * (1.0 - yv * yv)
        dy[:, :, : lay.dims[-1]] = g
        need_w, need_x = ctx.needs_input_grad[1], ctx.needs_input_grad[0]
        splits = default_splits(B) if need_w else 1
        ws = torch.empty(lay.bwd_ws_floats(B, splits), dtype=torch.float32, device=dev)
        grads = torch.empty_like(arena) if need_w else None
        dx = torch.empty((B, lay.ld_in), dtype=torch.float32, device=dev) if need_x else None
        if not (need_w or need_x):
            return None, None, None, None
        with torch.cuda.device(dev):
            L.check(L.lib.pqlk_mlp_backward(C.byref(lay.desc), L.ptr(arena), L.ptr(x_pad), x_pad.stride(0), B, L.ptr(acts),
                                            L.ptr(dy), L.ptr(grads), splits, L.ptr(dx), lay.ld_in if need_x else 0, 0, 0,
                                            None, 0, L.ptr(ws), ws.numel(), L.stream(dev)))
        return (dx[:, : ctx.in_cols] if need_x else None), grads, None, None


class FusedMLP(nn.Module):
    """n_nets structurally identical Linear->ELU->...->Linear nets in one arena."""

    key_prefixes = ("net.",)  # reference state_dict prefix per net

    def __init__(self, in_dim, out_dim, hidden_layers=None, n_nets=1, out_act=L.ACT_NONE):
        super().__init__()
        hidden = list(HIDDEN_DEFAULT if hidden_layers is None else hidden_layers)
        self.layout = ArenaLayout([_first(in_dim), *hidden, int(out_dim)], n_nets)
        self.out_act = out_act
        self.arena = nn.Parameter(torch.zeros(self.layout.total, dtype=torch.float32))
        self.reset_parameters()

    # ---- init = nn.Linear default (kaiming_uniform(a=sqrt(5)) -> U(+-1/sqrt(fan_in)) for W and b) -------
    @torch.no_grad()
    def reset_parameters(self):
        lay = self.layout
        self.arena.zero_()
        for n in range(lay.n_nets):
            for l in range(lay.n_layers):
                bound = 1.0 / (lay.dims[l] ** 0.5)
                lay.weight(self.arena.data, n, l).uniform_(-bound, bound)
                lay.bias(self.arena.data, n, l).uniform_(-bound, bound)

    @contextlib.contextmanager
    def leased(self):
        """Reads of this module's arena enqueued on the caller's CURRENT stream.  A *published* snapshot
        (pql_amd.utils.handoff.ArenaPublisher: what a learner's `start()` / `update()` hand out) is filled on the learner's
        own stream, and the host runs far ahead of the GPU: the caller's stream first waits for the snapshot's ready event,
        and the publisher may not refill the slot before the reads enqueued inside this block have run.  A plain module
        (no lease) follows the usual stream convention: nothing to do."""
        from pql_amd.utils import handoff as H
        lease = H.lease_of(self)
        if lease is None or not self.arena.is_cuda:
            yield
            return
        with H.LOCK:
            st = torch.cuda.current_stream(self.arena.device)
            if lease.ready is not None:
                st.wait_event(lease.ready)
            try:
                yield
            finally:
                H.release(lease, st)

    # ---- reference-keyed (de)serialisation ---------------------------------------------------------
    def state_dict(self, *args, destination=None, prefix="", keep_vars=False, **kw):
        out = OrderedDict() if destination is None else destination
        lay = self.layout
        with self.leased():   # the clones are enqueued behind the snapshot's ready event (torch.save / evaluator specs of a snapshot)
            for n, pre in enumerate(self.key_prefixes):
                for l in range(lay.n_layers):
                    out[f"{prefix}{pre}{2 * l}.weight"] = lay.weight(self.arena.data, n, l).clone()
                    out[f"{prefix}{pre}{2 * l}.bias"] = lay.bias(self.arena.data, n, l).clone()
        return out

    @torch.no_grad()
    def load_state_dict(self, state_dict, strict=True, assign=False):
        lay = self.layout
        missing = []
        for n, pre in enumerate(self.key_prefixes):
            for l in range(lay.n_layers):
                for kind, view in (("weight", lay.weight(self.arena.data, n, l)), ("bias", lay.bias(self.arena.data, n, l))):
                    key = f"{pre}{2 * l}.{kind}"
                    if key not in state_dict:
                        missing.append(key)
                        continue
                    src = torch.as_tensor(state_dict[key])
                    if tuple(src.shape) != tuple(view.shape):
                        raise RuntimeError(f"size mismatch for {key}: {tuple(src.shape)} vs {tuple(view.shape)}")
                    view.copy_(src.to(view.device, torch.float32))
        if strict and missing:
            raise RuntimeError(f"Missing key(s) in state_dict: {missing}")
        return nn.modules.module._IncompatibleKeys(missing, [])

    def named_views(self):
        """(reference key, strided view into the arena) pairs -- the per-tensor picture of the flat arena."""
        lay = self.layout
        for n, pre in enumerate(self.key_prefixes):
            for l in range(lay.n_layers):
                yield f"{pre}{2 * l}.weight", lay.weight(self.arena.data, n, l)
                yield f"{pre}{2 * l}.bias", lay.bias(self.arena.data, n, l)

    def num_params(self):
        return self.layout.count_params()

    def __deepcopy__(self, memo):
        """deepcopy of a *published* snapshot (pql_amd.utils.handoff.ArenaPublisher: what a learner's `update()` returns
        when it runs on its own stream) must not read the arena before the snapshot has been written, and must keep the
        publisher from overwriting it mid-copy: fence the copier's stream on both sides.  A plain module copies as usual."""
        from copy import deepcopy
        from pql_amd.utils import handoff as H
        lease = H.lease_of(self)
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        with H.LOCK:
            st = torch.cuda.current_stream(self.arena.device) if (lease is not None and self.arena.is_cuda) else None
            if st is not None and lease.ready is not None:
                st.wait_event(lease.ready)
            for k, v in self.__dict__.items():
                if k != "_pql_lease":
                    new.__dict__[k] = deepcopy(v, memo)
            if st is not None:
                H.release(lease, st)
        return new

    def _run(self, x):
        with self.leased():
            return FusedMlpFn.apply(x, self.arena, self.layout, self.out_act)


class MLPNet(FusedMLP):
    """mlp.py:27-40."""

    def __init__(self, in_dim, out_dim, hidden_layers=None, use_batchnorm=False):
        if use_batchnorm:
            raise NotImplementedError("BatchNorm MLP (CrossQ) is out of scope")
        super().__init__(in_dim, out_dim, hidden_layers, n_nets=1, out_act=L.ACT_NONE)
        self.init_kwargs = dict(in_dim=_first(in_dim), out_dim=int(out_dim), hidden_layers=self.layout.dims[1:-1])

    def forward(self, x):
        return self._run(x)[0]


class TanhMLPPolicy(FusedMLP):
    """mlp.py:177-179: tanh(MLP(state))."""

    def __init__(self, state_dim, act_dim, hidden_layers=None):
        super().__init__(state_dim, act_dim, hidden_layers, n_nets=1, out_act=L.ACT_TANH)
        self.init_kwargs = dict(state_dim=_first(state_dim), act_dim=int(act_dim), hidden_layers=self.layout.dims[1:-1])

    def forward(self, state):
        return self._run(state)[0]


class TanhDiagGaussianMLPPolicy(FusedMLP):
    """mlp.py:144-174: MLP -> [mu | log_std], a = tanh(mu + eps * exp(clamp(log_std, -5, 5))) (SquashedNormal), with the
    summed log-probability.  The head runs as one HIP launch (`pqlk_sg_head_forward`); the learner-side backward is
    `pqlk_sg_head_backward` (pql_amd/algo/sac.py).  Inference-only API here (no autograd through the sample)."""

    log_std_min, log_std_max = -5, 5

    def __init__(self, state_dim, act_dim, hidden_layers=None):
        self.act_dim = int(act_dim)
        super().__init__(state_dim, 2 * self.act_dim, hidden_layers, n_nets=1, out_act=L.ACT_NONE)
        self.init_kwargs = dict(state_dim=_first(state_dim), act_dim=self.act_dim, hidden_layers=self.layout.dims[1:-1])

    @torch.no_grad()
    def _head(self, state, eps, want_logp):
        L.require_gpu(self.arena, "parameter arena")
        x_pad = pad_cols(state.to(torch.float32), self.layout.ld_in)
        B, dev, A = x_pad.shape[0], x_pad.device, self.act_dim
        with self.leased():
            y = output_view(self.layout, mlp_forward_raw(self.layout, self.arena.data, x_pad, L.ACT_NONE), B)[0]
        act = torch.empty((B, A), dtype=torch.float32, device=dev)
        logp = torch.empty((B, 1), dtype=torch.float32, device=dev) if want_logp else None
        with torch.cuda.device(dev):
            L.check(L.lib.pqlk_sg_head_forward(L.ptr(y), y.stride(0), L.ptr(eps), B, A, L.ptr(act), A, L.ptr(logp), L.stream(dev)))
        return act, logp

    def forward(self, state, sample=False):
        return self.get_actions(state, sample=sample)

    def get_actions(self, state, sample=True, eps=None):
        """sample=True: rsample (eps ~ N(0,1) drawn here unless supplied); False: the distribution mean tanh(mu)."""
        if sample and eps is None:
            eps = torch.empty((state.shape[0], self.act_dim), dtype=torch.float32, device=state.device).normal_()
        return self._head(state, eps.contiguous() if sample else None, False)[0]

    def get_actions_logprob(self, state, eps=None):
        """-> (actions, None, log_prob (B, 1)); the middle slot is the reference's distribution object."""
        if eps is None:
            eps = torch.empty((state.shape[0], self.act_dim), dtype=torch.float32, device=state.device).normal_()
        act, logp = self._head(state, eps.contiguous(), True)
        return act, None, logp


class DoubleQ(FusedMLP):
    """mlp.py:186-203: twin Q(s,a) heads on cat(state, action)."""

    key_prefixes = ("net_q1.net.", "net_q2.net.")
    num_atoms = 1

    def __init__(self, state_dim, act_dim, hidden_layers=None, out_dim=1):
        self.state_dim, self.act_dim = _first(state_dim), int(act_dim)
        super().__init__(self.state_dim + self.act_dim, out_dim, hidden_layers, n_nets=2, out_act=L.ACT_NONE)
        # constructor arguments as plain data: lets a module cross a process boundary as (class, kwargs, state_dict)
        self.init_kwargs = dict(state_dim=self.state_dim, act_dim=self.act_dim, hidden_layers=self.layout.dims[1:-1], out_dim=int(out_dim))

    def _heads(self, state, action):
        return self._run(torch.cat((state, action), dim=1))

    def get_q1_q2(self, state, action):
        y = self._heads(state, action)
        return y[0], y[1]

    def get_q_min(self, state, action):
        return torch.min(*self.get_q1_q2(state, action))

    def get_q1(self, state, action):
        return self._heads(state, action)[0]


class DistributionalDoubleQ(DoubleQ):
    """mlp.py:244-267: twin categorical critics, softmax over `num_atoms` on a fixed support."""

    def __init__(self, state_dim, act_dim, v_min=-10, v_max=10, num_atoms=51, device="cuda", hidden_layers=None):
        super().__init__(state_dim, act_dim, hidden_layers, out_dim=num_atoms)
        self.device = device
        self.v_min, self.v_max, self.num_atoms = v_min, v_max, int(num_atoms)
        self.z_atoms = torch.linspace(v_min, v_max, num_atoms, device=device)  # plain attribute, as mlp.py:253
        self.init_kwargs = dict(state_dim=self.state_dim, act_dim=self.act_dim, v_min=v_min, v_max=v_max, num_atoms=self.num_atoms,
                                device=str(device), hidden_layers=self.layout.dims[1:-1])

    def get_q1_q2(self, state, action):
        y = self._heads(state, action)
        return torch.softmax(y[0], dim=1), torch.softmax(y[1], dim=1)

    def get_q_min(self, state, action):
        p1, p2 = self.get_q1_q2(state, action)
        z = self.z_atoms.to(p1.device)
        return torch.min(torch.sum(p1 * z, dim=1), torch.sum(p2 * z, dim=1))

    def get_q1(self, state, action):
        return torch.softmax(self._heads(state, action)[0], dim=1)
