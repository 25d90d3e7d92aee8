"""n-step transition assembler on the GPU.

Drop-in for `pql/replay/nstep_replay.py` (`NStepReplay` :6-71, `compute_nstep_return` :74-92): same
constructor and `add_to_buffer(obs, actions, rewards, next_obs, dones)` -> five tensors in the
reference's time-major row order.  The five `torch.cat` FIFO shifts per env-step become a circular
per-env window updated and emitted by one HIP launch per env-step (pqlk_nstep_push_emit).
"""
from __future__ import annotations

import ctypes as C

import torch

from pql_amd import _lib as L
from pql_amd.replay.simple_replay import _obs_width


class NStepReplay:
    def __init__(self, obs_dim, action_dim: int, num_envs: int = 1, nstep: int = 3, device="cuda", gamma: float = 0.99,
                 left_agent: bool = False):
        if left_agent:
            raise NotImplementedError("left_agent belongs to the bimanual fork variants, out of scope")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.PqlkError(f"NStepReplay must live on a GPU (got {device}); pql_amd has no CPU path")
        self.num_envs, self.nstep, self.gamma = int(num_envs), int(nstep), gamma
        self.O, self.A = _obs_width(obs_dim), int(action_dim)
        self.nstep_count = 0
        # gamma^i in python double, rounded once to fp32 (nstep_replay.py:24)
        self._gamma_pow = (C.c_float * self.nstep)(*[self.gamma ** i for i in range(self.nstep)])
        self.gamma_array = torch.tensor([self.gamma ** i for i in range(self.nstep)], device=self.device).view(-1, 1)
        self.win_ld = int(L.lib.pqlk_replay_rec_ld(self.O, self.A))
        self.window = torch.zeros((self.num_envs, self.nstep, self.win_ld), dtype=torch.float32, device=self.device)

    def rows_out(self, T):
        """Rows the next add_to_buffer call with horizon T emits."""
        if self.nstep <= 1:
            return T * self.num_envs
        return max(T - max(self.nstep - 1 - self.nstep_count, 0), 0) * self.num_envs

    @torch.no_grad()
    def add_to_buffer(self, obs, actions, rewards, next_obs, dones, out=None):
        """`out`: optional preallocated (M,O),(M,A),(M,1),(M,O),(M,1) fp32 destination (a producer that hands its blocks to
        other streams keeps them double-buffered instead of allocating per call)."""
        if self.nstep <= 1:  # pass-through (nstep_replay.py:66-67)
            if out is None:
                return obs, actions, rewards, next_obs, dones
            for dst, src in zip(out, (obs, actions, rewards, next_obs, dones)):
                dst.copy_(src.reshape(dst.shape))
            return out
        N, O, A = self.num_envs, self.O, self.A
        T = obs.shape[1]
        f = dict(dtype=torch.float32, device=self.device)
        obs = obs.reshape(N, T, O).to(**f).contiguous()
        actions = actions.reshape(N, T, A).to(**f).contiguous()
        rewards = rewards.reshape(N, T).to(**f).contiguous()
        next_obs = next_obs.reshape(N, T, O).to(**f).contiguous()
        dones = dones.reshape(N, T).to(**f).contiguous()
        first = max(self.nstep - 1 - self.nstep_count, 0)
        steps_out = max(T - first, 0)
        if steps_out == 0:
            # reference: torch.cat([]) raises when the first call is shorter than the window (:65)
            raise RuntimeError("NStepReplay.add_to_buffer: no complete n-step window yet (first call needs T >= nstep)")
        M = steps_out * N
        if out is None:
            out = (torch.empty((M, O), **f), torch.empty((M, A), **f), torch.empty((M, 1), **f), torch.empty((M, O), **f),
                   torch.empty((M, 1), **f))
        elif tuple(out[0].shape) != (M, O) or tuple(out[1].shape) != (M, A) or any(not t.is_contiguous() for t in out):
            raise L.PqlkError(f"NStepReplay.add_to_buffer: `out` must be contiguous blocks of {M} rows")
        rows = C.c_int64(0)
        with torch.cuda.device(self.device):
            L.check(L.lib.pqlk_nstep_push_emit(L.ptr(self.window), N, self.nstep, O, A, self.nstep_count, T,
                                               L.ptr(obs), L.ptr(actions), L.ptr(rewards), L.ptr(next_obs), L.ptr(dones),
                                               self._gamma_pow, *[L.ptr(t) for t in out], C.byref(rows),
                                               L.stream(self.device)))
        assert rows.value == M
        self.nstep_count += T
        return out
