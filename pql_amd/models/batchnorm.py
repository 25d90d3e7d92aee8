"""`DoubleQBatchNorm`: the CrossQ critic (reference pql/models/mlp.py:224-241 with `create_simple_mlp(use_batchnorm=True)`,
:15-24): twin `Linear -> BatchNorm1d -> ELU -> ... -> Linear` Q heads on `cat(state, action)`.

MI355X form: every Linear is a one-layer `PqlMlpDesc` call on the fp32-MFMA GEMMs (forward with the raw pre-activation
as output, backward giving dW / db / dX), the batch statistics one `pqlk_batch_moments` launch per (net, layer), and the
normalise + ELU pass, its backward and the column sums it needs are `pqlk_bn_elu_forward / _backward`
(pql_amd/csrc/bn.hip).  All parameters -- Linear weights and biases and the BatchNorm gamma / beta of both nets -- live in
ONE flat arena so that the optimiser's global-norm clip + AdamW is a single launch over it, like the other critics.
state_dict keys are the reference's (`net_q{1,2}.net.{0,3,6,9}.{weight,bias}` for the Linears,
`net_q{1,2}.net.{1,4,7}.{weight,bias,running_mean,running_var,num_batches_tracked}` for the norms).
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import torch
from torch import nn

from pql_amd import _lib as L
from pql_amd.models.mlp import HIDDEN_DEFAULT, ArenaLayout, _first, default_splits, pad_cols

BN_EPS, BN_MOMENTUM = 1e-5, 0.1   # nn.BatchNorm1d defaults


class DoubleQBatchNorm(nn.Module):
    key_prefixes = ("net_q1.net.", "net_q2.net.")
    num_atoms = 1

    def __init__(self, state_dim, act_dim, hidden_layers=None):
        super().__init__()
        self.state_dim, self.act_dim = _first(state_dim), int(act_dim)
        hidden = list(HIDDEN_DEFAULT if hidden_layers is None else hidden_layers)
        self.dims = [self.state_dim + self.act_dim, *hidden, 1]
        self.n_layers = len(self.dims) - 1
        self.init_kwargs = dict(state_dim=self.state_dim, act_dim=self.act_dim, hidden_layers=hidden)
        self.lin = [ArenaLayout([self.dims[l], self.dims[l + 1]], 1) for l in range(self.n_layers)]   # one-layer descriptors
        # flat arena: per net, per layer: [W (out, ld(in)) | b (ld(out))] then, for hidden layers, [gamma (ld(out)) | beta (ld(out))]
        self.off = {}
        o = 0
        for n in range(2):
            for l in range(self.n_layers):
                self.off[(n, l, "lin")] = o
                o += self.lin[l].total
                if l < self.n_layers - 1:
                    w = L.ld(self.dims[l + 1])
                    self.off[(n, l, "gamma")], self.off[(n, l, "beta")] = o, o + w
                    o += 2 * w
        self.total = o
        self.arena = nn.Parameter(torch.zeros(self.total, dtype=torch.float32))
        # running statistics: per net, per hidden layer [running_mean (ld) | running_var (ld)]
        self.soff = {}
        o = 0
        for n in range(2):
            for l in range(self.n_layers - 1):
                w = L.ld(self.dims[l + 1])
                self.soff[(n, l)] = (o, o + w)
                o += 2 * w
        self.register_buffer("stats", torch.zeros(o, dtype=torch.float32))
        self.register_buffer("num_batches_tracked", torch.zeros(2 * (self.n_layers - 1), dtype=torch.int64))
        self.reset_parameters()
        self._ws = {}

    # ---- parameter views -------------------------------------------------------------------------------------
    def weight(self, n, l):
        return self.lin[l].weight(self.arena.data[self.off[(n, l, "lin")]:], 0, 0)

    def bias(self, n, l):
        return self.lin[l].bias(self.arena.data[self.off[(n, l, "lin")]:], 0, 0)

    def bn_param(self, n, l, which, arena=None):
        a = self.arena.data if arena is None else arena
        o = self.off[(n, l, which)]
        return a[o: o + self.dims[l + 1]]

    def running(self, n, l, which):
        o = self.soff[(n, l)][0 if which == "mean" else 1]
        return self.stats[o: o + self.dims[l + 1]]

    @torch.no_grad()
    def reset_parameters(self):
        self.arena.zero_()
        for n in range(2):
            for l in range(self.n_layers):
                bound = 1.0 / (self.dims[l] ** 0.5)    # nn.Linear default
                self.weight(n, l).uniform_(-bound, bound)
                self.bias(n, l).uniform_(-bound, bound)
                if l < self.n_layers - 1:
                    self.bn_param(n, l, "gamma").fill_(1.0)
                    self.running(n, l, "var").fill_(1.0)

    def named_views(self):
        """(reference key, view) pairs of the trainable tensors."""
        for n, pre in enumerate(self.key_prefixes):
            for l in range(self.n_layers):
                yield f"{pre}{3 * l}.weight", self.weight(n, l)
                yield f"{pre}{3 * l}.bias", self.bias(n, l)
                if l < self.n_layers - 1:
                    yield f"{pre}{3 * l + 1}.weight", self.bn_param(n, l, "gamma")
                    yield f"{pre}{3 * l + 1}.bias", self.bn_param(n, l, "beta")

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False, **kw):
        out = OrderedDict() if destination is None else destination
        for k, v in self.named_views():
            out[prefix + k] = v.detach().clone()
        for n, pre in enumerate(self.key_prefixes):
            for l in range(self.n_layers - 1):
                out[f"{prefix}{pre}{3 * l + 1}.running_mean"] = self.running(n, l, "mean").clone()
                out[f"{prefix}{pre}{3 * l + 1}.running_var"] = self.running(n, l, "var").clone()
                out[f"{prefix}{pre}{3 * l + 1}.num_batches_tracked"] = self.num_batches_tracked[n * (self.n_layers - 1) + l].clone()
        return out

    @torch.no_grad()
    def load_state_dict(self, state_dict, strict=True, assign=False):
        missing = []
        for k, v in self.named_views():
            if k in state_dict:
                v.copy_(torch.as_tensor(state_dict[k]).to(v.device, torch.float32))
            else:
                missing.append(k)
        for n, pre in enumerate(self.key_prefixes):
            for l in range(self.n_layers - 1):
                for which in ("mean", "var"):
                    k = f"{pre}{3 * l + 1}.running_{which}"
                    if k in state_dict:
                        self.running(n, l, which).copy_(torch.as_tensor(state_dict[k]).to(self.stats.device, torch.float32))
                    else:
                        missing.append(k)
                k = f"{pre}{3 * l + 1}.num_batches_tracked"
                if k in state_dict:
                    self.num_batches_tracked[n * (self.n_layers - 1) + l] = int(state_dict[k])
        if strict and missing:
            raise RuntimeError(f"Missing key(s) in state_dict: {missing}")
        return nn.modules.module._IncompatibleKeys(missing, [])

    def num_params(self):
        return sum(v.numel() for _, v in self.named_views())

    # ---- raw launch sequences --------------------------------------------------------------------------------
    def _workspace(self, M, dev):
        ws = self._ws.get(M)
        if ws is not None and ws["dev"] == dev:
            return ws
        f = dict(dtype=torch.float32, device=dev)
        ws = dict(dev=dev, z={}, y={}, mean={}, var={}, splits=default_splits(M))
        wmax = max(L.ld(d) for d in self.dims[1:])
        for n in range(2):
            for l in range(self.n_layers):
                w = L.ld(self.dims[l + 1])
                ws["z"][(n, l)] = torch.zeros((M, w), **f)
                if l < self.n_layers - 1:
                    ws["y"][(n, l)] = torch.zeros((M, w), **f)      # pad columns stay zero
                    ws["mean"][(n, l)] = torch.zeros(w, **f)
                    ws["var"][(n, l)] = torch.ones(w, **f)
        ws["mom_scratch"] = torch.zeros(64 * wmax * 3, **f)
        ws["bn_scratch"] = torch.zeros(128 * wmax, **f)
        ws["dcur"] = [torch.zeros((M, wmax), **f) for _ in range(2)]
        ws["dx0"] = [torch.zeros((M, L.ld(self.dims[0])), **f) for _ in range(2)]
        ws["bwd"] = torch.empty(max(lay.bwd_ws_floats(M, ws["splits"]) for lay in self.lin), **f)
        self._ws[M] = ws
        return ws

    @torch.no_grad()
    def forward_raw(self, x_pad, training=True):
        """x_pad (M, ld(in)) with zero pad columns -> Q (2, M, 32) (column 0).  Keeps z / y / batch statistics of every layer
        for `backward_raw`; training=True normalises with batch statistics and updates the running ones (momentum 0.1)."""
        L.require_gpu(self.arena, "parameter arena")
        M, dev = x_pad.shape[0], x_pad.device
        ws = self._workspace(M, dev)
        arena = self.arena.data
        with torch.cuda.device(dev):
            st = L.stream(dev)
            for n in range(2):
                x, ldx = x_pad, x_pad.stride(0)
                for l in range(self.n_layers):
                    lay, z = self.lin[l], ws["z"][(n, l)]
                    L.check(L.lib.pqlk_mlp_forward(C.byref(lay.desc), L.ptr(arena[self.off[(n, l, "lin")]:]), None, 1, L.ptr(x), ldx, M,
                                                   L.ACT_NONE, None, 0.0, 0.0, L.ptr(z), None, 0, st))
                    if l == self.n_layers - 1:
                        break
                    cols, w = self.dims[l + 1], z.stride(0)
                    mean, var = ws["mean"][(n, l)], ws["var"][(n, l)]
                    if training:
                        L.check(L.lib.pqlk_batch_moments(L.ptr(z), w, M, cols, L.ptr(mean), L.ptr(var), L.ptr(ws["mom_scratch"]), st))
                    rm, rv = self.running(n, l, "mean"), self.running(n, l, "var")
                    y = ws["y"][(n, l)]
                    L.check(L.lib.pqlk_bn_elu_forward(L.ptr(z), w, M, cols, L.ptr(mean), L.ptr(var), L.ptr(self.bn_param(n, l, "gamma")),
                                                      L.ptr(self.bn_param(n, l, "beta")), BN_EPS, 1 if training else 0, BN_MOMENTUM,
                                                      L.ptr(rm), L.ptr(rv), L.ptr(y), st))
                    x, ldx = y, w
            if training:
                self.num_batches_tracked += 1
        return torch.stack((ws["z"][(0, self.n_layers - 1)], ws["z"][(1, self.n_layers - 1)]))

    @torch.no_grad()
    def backward_raw(self, x_pad, dq, grads=None, need_dx=False):
        """Backward of the LAST training-mode `forward_raw` on the same x_pad.  dq (2, M, 32): d loss / d Q (column 0).
        grads: flat tensor like the arena, overwritten with the parameter gradient (None: parameters frozen).
        Returns d loss / d x summed over the nets, (M, ld(in)), when need_dx."""
        M, dev = x_pad.shape[0], x_pad.device
        ws = self._workspace(M, dev)
        arena = self.arena.data
        splits = ws["splits"] if grads is not None else 1
        with torch.cuda.device(dev):
            st = L.stream(dev)
            for n in range(2):
                dcur, ld_d = dq[n], dq.stride(1)
                for l in range(self.n_layers - 1, -1, -1):
                    lay = self.lin[l]
                    if l < self.n_layers - 1:   # through ELU and BatchNorm: dcur (grad of y_l) -> dz_l, in place
                        cols, w = self.dims[l + 1], ws["z"][(n, l)].stride(0)
                        gg = self.bn_param(n, l, "gamma", grads) if grads is not None else None
                        gb = self.bn_param(n, l, "beta", grads) if grads is not None else None
                        L.check(L.lib.pqlk_bn_elu_backward(L.ptr(dcur), L.ptr(ws["y"][(n, l)]), L.ptr(ws["z"][(n, l)]), w, M, cols,
                                                           L.ptr(ws["mean"][(n, l)]), L.ptr(ws["var"][(n, l)]),
                                                           L.ptr(self.bn_param(n, l, "gamma")), BN_EPS, L.ptr(dcur), L.ptr(gg), L.ptr(gb),
                                                           L.ptr(ws["bn_scratch"]), st))
                    x_in = x_pad if l == 0 else ws["y"][(n, l - 1)]
                    if l > 0:      # grad of y_{l-1}: ping-pong buffers, viewed with that layer's row stride
                        w_in = x_in.stride(0)
                        dx = ws["dcur"][l & 1].view(-1)[: M * w_in].view(M, w_in)
                    else:
                        dx = ws["dx0"][n] if need_dx else None
                    g_lin = grads[self.off[(n, l, "lin")]:] if grads is not None else None
                    L.check(L.lib.pqlk_mlp_backward(C.byref(lay.desc), L.ptr(arena[self.off[(n, l, "lin")]:]), L.ptr(x_in), x_in.stride(0), M,
                                                    L.ptr(ws["z"][(n, l)]), L.ptr(dcur), L.ptr(g_lin), splits,
                                                    L.ptr(dx), dx.stride(0) if dx is not None else 0, 0, 0, None, 0,
                                                    L.ptr(ws["bwd"]), ws["bwd"].numel(), st))
                    dcur = dx
            if need_dx:
                return ws["dx0"][0] + ws["dx0"][1]
        return None

    # ---- reference surface (inference-style calls; train()/eval() select batch vs running statistics) -----------
    def _heads(self, state, action):
        x = pad_cols(torch.cat((state, action), dim=1).to(torch.float32), L.ld(self.dims[0]))
        return self.forward_raw(x, training=self.training)

    def get_q1_q2(self, state, action):
        q = self._heads(state, action)
        return q[0, :, :1].clone(), q[1, :, :1].clone()

    def get_q_min(self, state, action):
        return torch.min(*self.get_q1_q2(state, action))

    def get_q1(self, state, action):
        return self._heads(state, action)[0, :, :1].clone()
