"""Synthetic vectorised environment: the Isaac-Gym rollout stub named by BASELINE.json.

Honours the env contract the reference consumes (pql/utils/isaacgym_util.py + wrappers):
`reset() -> obs (N, O)`, `step(a) -> (obs, reward (N,), done (N,), info)`, attributes
`observation_space.shape`, `action_space.shape`, `max_episode_length`, `num_envs`.

Transitions are a pure function of (seed, global env id, step counter) -- a counter-based generator -- so
any data-parallel sharding of the env axis reproduces the same per-env streams:
obs ~ N(0,1), reward ~ N(0,1) (+ a small action-dependent term so the critic has signal),
done ~ Bernoulli(1/episode_length), info['TimeLimit.truncated'] = False.
"""
from __future__ import annotations

import math
import os
from types import SimpleNamespace

import torch

TASK_SHAPES = dict(AllegroHand=(88, 16), ShadowHand=(211, 20), Humanoid=(108, 21), Ant=(60, 8), Anymal=(48, 12), Toy=(8, 2))

_M = 0xFFFFFFFF


def _hash32(x):
    """xorshift-multiply avalanche on int64 tensors holding 32-bit values."""
    x = x & _M
    x = ((x ^ (x >> 16)) * 0x7FEB352D) & _M
    x = ((x ^ (x >> 15)) * 0x846CA68B) & _M
    return x ^ (x >> 16)


class SyntheticVecEnv:
    def __init__(self, num_envs, obs_dim, act_dim, device="cuda", seed=42, episode_length=300, env_offset=0):
        self.num_envs, self.obs_dim, self.act_dim = int(num_envs), int(obs_dim), int(act_dim)
        self.device = torch.device(device)
        self.seed = int(seed)
        self.max_episode_length = int(episode_length)
        self.observation_space = SimpleNamespace(shape=(self.obs_dim,))
        self.action_space = SimpleNamespace(shape=(self.act_dim,))
        self.t = 0
        self.env_ids = (torch.arange(self.num_envs, device=self.device, dtype=torch.int64) + int(env_offset))
        self._p_done = 1.0 / float(episode_length)
        self.env_offset = int(env_offset)
        self._no_trunc = None
        self._obs = None

    def _uniform(self, stream, width):
        """(N, width) uniforms in (0,1): hash of (seed, env id, step, stream, column)."""
        col = torch.arange(width, device=self.device, dtype=torch.int64)
        key = _hash32(self.env_ids * 0x9E3779B1 + self.seed * 0x85EBCA77 + self.t * 0xC2B2AE3D + stream * 0x27D4EB2F)
        h = _hash32(key.unsqueeze(1) * 0x165667B1 + col.unsqueeze(0) * 0x9E3779B1 + 0x5BD1E995)
        return (h.to(torch.float32) + 0.5) * (1.0 / 4294967296.0)

    def _normal(self, stream, width):
        u1, u2 = self._uniform(2 * stream, width), self._uniform(2 * stream + 1, width)
        return torch.sqrt(-2.0 * torch.log(u1)) * torch.cos((2.0 * math.pi) * u2)

    @torch.no_grad()
    def reset(self):
        self.t = 0
        self._obs = self._normal(1, self.obs_dim)
        return self._obs

    @torch.no_grad()
    def step(self, action):
        self.t += 1
        if self.device.type == "cuda" and not os.environ.get("PQL_SYNTH_TORCH"):   # one HIP launch instead of ~150 elementwise torch launches
            return self._step_hip(action)
        return self._step_torch(action)

    def _step_torch(self, action):
        next_obs = self._normal(1, self.obs_dim)
        reward = self._normal(2, 1).squeeze(1) - 0.1 * (action * action).mean(dim=1)
        done = self._uniform(7, 1).squeeze(1) < self._p_done
        info = {"TimeLimit.truncated": torch.zeros_like(done)}
        self._obs = next_obs
        return next_obs, reward, done, info

    def _step_hip(self, action):
        """Same transition as `_step_torch`, one launch (`pqlk_synth_env_step`, include/pqlk.h)."""
        from pql_amd import _lib as L
        n, dev = self.num_envs, self.device
        next_obs = torch.empty((n, self.obs_dim), dtype=torch.float32, device=dev)
        reward = torch.empty(n, dtype=torch.float32, device=dev)
        done = torch.empty(n, dtype=torch.bool, device=dev)
        act = action.to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            L.check(L.lib.pqlk_synth_env_step(n, self.obs_dim, self.act_dim, self.seed & 0xFFFFFFFF, self.env_offset & 0xFFFFFFFF,
                                              self.t & 0xFFFFFFFF, float(self._p_done), L.ptr(act), L.ptr(next_obs), L.ptr(reward),
                                              L.ptr(done), L.stream(dev)))
        if self._no_trunc is None:
            self._no_trunc = torch.zeros(n, dtype=torch.bool, device=dev)
        self._obs = next_obs
        return next_obs, reward, done, {"TimeLimit.truncated": self._no_trunc}


def create_task_env(cfg, num_envs=None, env_offset=0):
    """Stand-in for pql.utils.isaacgym_util.create_task_env (:8-24)."""
    task = cfg.task
    name = task.name if task is not None else "AllegroHand"
    O, A = TASK_SHAPES.get(name, (88, 16))
    if task is not None:
        O = int(task.obs_dim) if task.get("obs_dim") else O
        A = int(task.act_dim) if task.get("act_dim") else A
    ep = int(task.get("episode_length") or 300) if task is not None else 300
    return SyntheticVecEnv(num_envs or cfg.num_envs, O, A, device=cfg.sim_device, seed=cfg.seed, episode_length=ep,
                           env_offset=env_offset)
