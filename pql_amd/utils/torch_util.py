"""`soft_update` and `RunningMeanStd` with the reference's names (pql/utils/torch_util.py:9-12, :68-114)."""
from __future__ import annotations


import torch

from pql_amd import _lib as L


@torch.no_grad()
def soft_update(target_net, current_net, tau: float):
    """theta' <- tau*theta + (1-tau)*theta'.  One launch over the flat arenas when both nets are
    pql_amd FusedMLP modules; any other nn.Module pair is rejected loudly (no eager fallback)."""
    ta, ca = getattr(target_net, "arena", None), getattr(current_net, "arena", None)
    if ta is None or ca is None:
        raise L.PqlkError("soft_update needs pql_amd models (flat parameter arenas)")
    with torch.cuda.device(ta.device):
        L.check(L.lib.pqlk_polyak(L.ptr(ta.data), L.ptr(ca.data), ta.numel(), float(tau), L.stream(ta.device)))


class RunningMeanStd:
    """Running per-feature mean/variance with the parallel (Chan) merge; count starts at epsilon.
    Batch moments come from one HIP launch (pqlk_batch_moments); the O(obs_dim) merge is torch."""

    def __init__(self, epsilon=1e-4, shape=(), device="cuda"):
        self.device = torch.device(device)
        self.mean = torch.zeros(shape, device=self.device)
        self.var = torch.ones(shape, device=self.device)
        self.epsilon = epsilon
        self.count = epsilon

    def batch_moments(self, x):
        x = x.reshape(x.shape[0], -1)
        cols = x.shape[1]
        bm = torch.empty(cols, dtype=torch.float32, device=x.device)
        bv = torch.empty(cols, dtype=torch.float32, device=x.device)
        x = x.contiguous()
        if getattr(self, "_scratch", None) is None or self._scratch.numel() < 64 * cols * 3 or self._scratch.device != x.device:
            self._scratch = torch.empty(64 * cols * 3, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            L.check(L.lib.pqlk_batch_moments(L.ptr(x), x.stride(0), x.shape[0], cols, L.ptr(bm), L.ptr(bv),
                                             L.ptr(self._scratch), L.stream(x.device)))
        return bm.view(self.mean.shape), bv.view(self.mean.shape)

    def update(self, x):
        bm, bv = self.batch_moments(x)
        self.merge_batch(bm, bv, x.shape[0])

    def merge_batch(self, bm, bv, n):
        """Fold one batch's moments in.  Data parallel (self.pg set): every rank folds in EVERY rank's batch
        moments, in rank order, so all replicas hold identical statistics; one all-gather of
        2*obs_dim floats per env step.  Device-agnostic (plain torch + torch.distributed)."""
        pg = getattr(self, "pg", None)
        if pg is None:
            self.update_from_moments(bm, bv, n)
            return
        world = torch.distributed.get_world_size(pg)
        packed = torch.cat([bm.reshape(-1), bv.reshape(-1)])
        if torch.distributed.get_backend(pg) == "gloo" and packed.is_cuda:   # rehearsal path: stage through the host
            hp = packed.cpu()
            hg = torch.empty(world * hp.numel(), dtype=hp.dtype)
            torch.distributed.all_gather_into_tensor(hg, hp, group=pg)
            gathered = hg.to(packed.device)
        else:
            gathered = torch.empty(world * packed.numel(), dtype=packed.dtype, device=packed.device)
            torch.distributed.all_gather_into_tensor(gathered, packed, group=pg)
        gathered = gathered.view(world, 2, -1)
        for r in range(world):
            self.update_from_moments(gathered[r, 0].view(self.mean.shape), gathered[r, 1].view(self.mean.shape), n)

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        tot = self.count + batch_count
        if (self.mean.is_cuda and self.mean.dtype == torch.float32 and batch_mean.is_cuda and batch_mean.dtype == torch.float32
                and batch_var.dtype == torch.float32 and self.mean.is_contiguous() and self.var.is_contiguous()):
            # one launch instead of twelve; new tensors, like the torch expression below: what get_states() handed out stays a snapshot
            bm, bv = batch_mean.contiguous(), batch_var.contiguous()
            mean, var = torch.empty_like(self.mean), torch.empty_like(self.var)
            with torch.cuda.device(self.mean.device):
                L.check(L.lib.pqlk_rms_merge(L.ptr(self.mean), L.ptr(self.var), L.ptr(bm), L.ptr(bv), float(self.count),
                                             float(batch_count), float(tot), self.mean.numel(), L.ptr(mean), L.ptr(var),
                                             L.stream(self.mean.device)))
            self.mean, self.var, self.count = mean, var, tot
            return
        delta = batch_mean - self.mean
        m2 = self.var * self.count + batch_var * batch_count + delta ** 2 * self.count * batch_count / tot
        self.mean = self.mean + delta * batch_count / tot
        self.var = m2 / tot
        self.count = tot

    def normalize(self, x, out=None):
        """Actor side: no clamp (torch_util.py:83-85).  `out`: optional (rows, >= cols) fp32 matrix to write into (row stride
        `out.stride(0)`, columns past `cols` untouched) -- the policy's zero-padded input tile."""
        if (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and self.mean.is_cuda and x.dim() >= 1
                and x.shape[-1] == self.mean.numel() and self.mean.dim() == 1 and x.numel() > 0):
            dst = torch.empty_like(x) if out is None else out
            with torch.cuda.device(x.device):
                L.check(L.lib.pqlk_rms_normalize(L.ptr(x), x.numel() // x.shape[-1], x.shape[-1], L.ptr(self.mean), L.ptr(self.var),
                                                 float(self.epsilon), L.ptr(dst), x.shape[-1] if out is None else out.stride(0),
                                                 L.stream(x.device)))
            return dst
        if out is not None:
            raise L.PqlkError("RunningMeanStd.normalize(out=...) needs contiguous fp32 GPU tensors")
        return (x - self.mean) / torch.sqrt(self.var + self.epsilon)

    def unnormalize(self, x):
        return x * torch.sqrt(self.var + self.epsilon) + self.mean

    def get_states(self, device=None):
        if device is not None:
            return self.mean.to(device), self.var.to(device), self.epsilon
        return self.mean, self.var, self.epsilon

    def load_state_dict(self, info):
        self.mean, self.var, self.count = info[0], info[1], info[2]
