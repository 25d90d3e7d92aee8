"""Replay ring buffer on HBM.

Drop-in for the reference's `pql/replay/simple_replay.py` (`create_buffer` :4-18, `ReplayBuffer` :21-104):
same constructor, `add_to_buffer(trajectory)`, `sample_batch(batch_size, device)`, public pointer
attributes `next_p / if_full / cur_capacity / capacity` and `buf_*` tensors.

MI355X design: the five SoA tensors of the reference become one array of fixed-stride, 128-byte
aligned records (layout in include/pqlk.h) so that a uniform random sample touches the minimum number
of HBM lines; `buf_obs`, `buf_action`, `buf_next_obs`, `buf_reward` are strided views into it and
`buf_done` a bool view computed on access.  Insert and gather are single HIP launches.
"""
from __future__ import annotations

import ctypes as C

import torch

from pql_amd import _lib as L


def _obs_width(obs_dim) -> int:
    if isinstance(obs_dim, int):
        return obs_dim
    if len(obs_dim) != 1:
        raise NotImplementedError("only flat observations are supported (the reference flattens them too)")
    return int(obs_dim[0])


def ring_plan(next_p: int, if_full: bool, capacity: int, m: int):
    """Integer pointer law of simple_replay.py:52-83 -> ordered (dst_start, src_start, length) segments."""
    p = next_p + m
    segs = []
    if p > capacity:
        if_full = True
        head = capacity - next_p
        if head > 0:
            segs.append((next_p, 0, head))
        p -= capacity
        if p > capacity:
            raise RuntimeError(f"cannot insert {m} rows into a ring of capacity {capacity} at pointer {next_p}")
        segs.append((0, m - p, p))  # the LAST p rows wrap to the front (simple_replay.py:66)
    else:
        segs.append((next_p, 0, m))
    return segs, p, if_full, (capacity if if_full else p)


class RecordRing:
    """Device record array + descriptor shared by ReplayBuffer (A >= 0) and the P-learner obs ring (A = -1)."""

    def __init__(self, capacity: int, obs_dim: int, act_dim: int, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.PqlkError(f"replay ring must live on a GPU (got {device}); pql_amd has no CPU path")
        self.capacity, self.O, self.A = int(capacity), int(obs_dim), int(act_dim)
        self.rec_ld = int(L.lib.pqlk_replay_rec_ld(self.O, self.A))
        self.records = torch.zeros((self.capacity, self.rec_ld), dtype=torch.float32, device=self.device)
        self.desc = L.PqlReplayDesc(self.records.data_ptr(), self.capacity, self.O, self.A, self.rec_ld, 0)
        o4 = (self.O + 3) & ~3
        self.off_nobs, self.off_act = o4, 2 * o4
        self.off_rd = 2 * o4 + ((max(self.A, 0) + 3) & ~3)
        self.version = 0   # bumped by every insert: the learners' draws-ahead tiles are stamped with it (stale tiles are re-gathered)

    def insert_segments(self, segs, obs, act=None, rew=None, nobs=None, done=None):
        self.version += 1
        with torch.cuda.device(self.device):
            st = L.stream(self.device)
            for dst, src, n in segs:
                if n == 0:
                    continue
                L.check(L.lib.pqlk_replay_insert(
                    C.byref(self.desc), dst, n,
                    L.ptr(obs[src:]), obs.stride(0),
                    L.ptr(act[src:]) if act is not None else None, act.stride(0) if act is not None else 0,
                    L.ptr(rew[src:]) if rew is not None else None, rew.stride(0) if rew is not None else 0,
                    L.ptr(nobs[src:]) if nobs is not None else None, nobs.stride(0) if nobs is not None else 0,
                    L.ptr(done[src:]) if done is not None else None, done.stride(0) if done is not None else 0,
                    st))


def create_buffer(capacity, obs_dim, action_dim, device="cuda", reserve_space=False):
    """Reference-shaped allocator (simple_replay.py:4-18): returns (obs, action, next_obs, reward, done).
    Kept for callers that want plain SoA tensors; ReplayBuffer itself uses the record layout."""
    if reserve_space:
        raise NotImplementedError("reserve_space (fp16 host-side obs) is not used by PQL and not supported")
    cap = (capacity,) if isinstance(capacity, int) else tuple(capacity)
    O = _obs_width(obs_dim)
    f = dict(dtype=torch.float32, device=device)
    return (torch.empty((*cap, O), **f), torch.empty((*cap, int(action_dim)), **f), torch.empty((*cap, O), **f),
            torch.empty((*cap, 1), **f), torch.empty((*cap, 1), dtype=torch.bool, device=device))


class ReplayBuffer:
    def __init__(self, capacity: int, obs_dim, action_dim: int, device="cuda", left_agent: bool = False,
                 reserve_space: bool = False):
        if left_agent or reserve_space:
            raise NotImplementedError("left_agent / reserve_space belong to the bimanual fork variants, out of scope")
        self.obs_dim = (obs_dim,) if isinstance(obs_dim, int) else tuple(obs_dim)
        self.action_dim = int(action_dim)
        self.device = torch.device(device)
        self.next_p = 0
        self.if_full = False
        self.cur_capacity = 0
        self.capacity = int(capacity)
        self.ring = RecordRing(self.capacity, _obs_width(obs_dim), self.action_dim, self.device)

    # ---- reference-named views -------------------------------------------------------------
    @property
    def records(self):
        return self.ring.records

    @property
    def buf_obs(self):
        return self.ring.records[:, : self.ring.O]

    @property
    def buf_next_obs(self):
        return self.ring.records[:, self.ring.off_nobs: self.ring.off_nobs + self.ring.O]

    @property
    def buf_action(self):
        return self.ring.records[:, self.ring.off_act: self.ring.off_act + self.ring.A]

    @property
    def buf_reward(self):
        return self.ring.records[:, self.ring.off_rd: self.ring.off_rd + 1]

    @property
    def buf_done(self):
        return self.ring.records[:, self.ring.off_rd + 1: self.ring.off_rd + 2] != 0

    # ---- a3 ---------------------------------------------------------------------------------
    @torch.no_grad()
    def add_to_buffer(self, trajectory):
        obs, actions, rewards, next_obs, dones = trajectory
        O, A = self.ring.O, self.ring.A
        f = dict(dtype=torch.float32, device=self.device)
        obs = obs.reshape(-1, O).to(**f).contiguous()
        actions = actions.reshape(-1, A).to(**f).contiguous()
        rewards = rewards.reshape(-1, 1).to(**f).contiguous()
        next_obs = next_obs.reshape(-1, O).to(**f).contiguous()
        dones = dones.reshape(-1, 1).to(**f).contiguous()
        segs, self.next_p, self.if_full, self.cur_capacity = ring_plan(self.next_p, self.if_full, self.capacity,
                                                                       rewards.shape[0])
        self.ring.insert_segments(segs, obs, actions, rewards, next_obs, dones)

    # ---- a4 ---------------------------------------------------------------------------------
    def draw_indices(self, batch_size, device=None):
        """The one RNG draw of sample_batch (simple_replay.py:87): same call, same shape/dtype/device."""
        return torch.randint(self.cur_capacity, size=(batch_size,), device=device or self.device)

    @torch.no_grad()
    def sample_batch(self, batch_size, device="cuda", indices=None):
        dev = self.device
        idx = self.draw_indices(batch_size) if indices is None else indices.to(device=dev, dtype=torch.int64).contiguous()
        B, O, A = idx.shape[0], self.ring.O, self.ring.A
        f = dict(dtype=torch.float32, device=dev)
        out = (torch.empty((B, O), **f), torch.empty((B, A), **f), torch.empty((B, 1), **f), torch.empty((B, O), **f),
               torch.empty((B, 1), **f))
        with torch.cuda.device(dev):
            L.check(L.lib.pqlk_replay_gather(C.byref(self.ring.desc), L.ptr(idx), B, *[L.ptr(t) for t in out],
                                             L.stream(dev)))
        tgt = torch.device(device)
        if tgt.type == "cuda" and tgt.index is None:
            tgt = dev
        return tuple(t.to(tgt) for t in out)
