"""Rollout side of PQL: steps the vectorised env, maintains the running observation statistics and the
n-step assembler, and hands transition blocks to the learners.

Drop-in for `pql/algo/pql_actor.py` (`PQLActor(env, cfg)`: `reset_agent`, `explore_env(env, timesteps, random)`
-> `(act_data, cri_data, steps)`, `obs_rms.get_states`, `return_tracker` / `step_tracker`, `add_info_tracker_log`).
Differences that matter on MI355X: no per-step host sync (episode trackers are device rings read at log time; the
reference's `torch.where(done)[0]` + `.tolist()` (:129-135) stalls the stream every env step), batch moments and the
n-step window run as single HIP launches, and data parallel ranks index the GLOBAL env axis for the mixed noise.
"""
from __future__ import annotations

import torch

from pql_amd.replay.nstep_replay import NStepReplay
from pql_amd.utils.common import handle_timeout
from pql_amd.utils.noise import add_mixed_normal_noise, add_normal_noise
from pql_amd.utils.schedule_util import ExponentialSchedule, LinearSchedule
from pql_amd.utils.torch_util import RunningMeanStd


class DeviceTracker:
    """Moving window over the last `max_len` finished-episode values, kept on the GPU (zero-filled like
    common.Tracker).  update(values, mask) scatters the masked values without a host sync."""

    def __init__(self, max_len, device):
        self.max_len = int(max_len)
        self.ring = torch.zeros(self.max_len + 1, device=device)   # last slot = discard bin
        self.ptr = torch.zeros((), dtype=torch.int64, device=device)

    def update(self, values, mask):
        pos = (self.ptr + torch.cumsum(mask.to(torch.int64), 0) - 1) % self.max_len
        self.ring.scatter_(0, torch.where(mask, pos, torch.full_like(pos, self.max_len)), values)
        self.ptr = (self.ptr + mask.sum()) % self.max_len

    def mean(self):
        return float(self.ring[: self.max_len].mean())


class PQLActor:
    def __init__(self, env, cfg, env_offset=0, total_envs=None):
        self.env, self.cfg = env, cfg
        self.obs_dim = env.observation_space.shape
        self.action_dim = env.action_space.shape[0]
        self.sim_device = torch.device(f"{cfg.sim_device}")
        self.v_learner_device = torch.device(f"cuda:{cfg.algo.v_learner_gpu}")
        self.p_learner_device = torch.device(f"cuda:{cfg.algo.p_learner_gpu}")
        self.env_offset, self.total_envs = int(env_offset), total_envs   # position on the GLOBAL env axis (data parallel)
        self.actor = None   # rollout replica of the policy, assigned by the driver
        self.obs = None
        if cfg.info_track_keys is not None:
            raise NotImplementedError("info_track_keys needs a simulator's info dict; out of scope")
        algo, n, dev = cfg.algo, cfg.num_envs, self.sim_device
        self.return_tracker = DeviceTracker(algo.tracker_len, dev)
        self.step_tracker = DeviceTracker(algo.tracker_len, dev)
        self.current_returns = torch.zeros(n, dtype=torch.float32, device=dev)
        self.current_lengths = torch.zeros(n, dtype=torch.float32, device=dev)
        self.obs_rms = RunningMeanStd(shape=self.obs_dim, device=dev) if algo.obs_norm else None
        self.n_step_buffer = NStepReplay(self.obs_dim, self.action_dim, n, algo.nstep, device=dev)
        self.noise_scheduler = self._make_scheduler(algo.noise)
        self._slabs = {}   # (N, T, .) trajectory slabs, allocated once per horizon length and reused

    @staticmethod
    def _make_scheduler(noise):
        if noise.decay == "linear":
            return LinearSchedule(noise.std_max, noise.std_min, noise.lin_decay_iters)
        if noise.decay == "exp":
            return ExponentialSchedule(noise.std_max, noise.exp_decay_rate, noise.std_min)
        return None

    # ---- small API kept from the reference ---------------------------------------------------
    def reset_agent(self):
        self.obs = self.env.reset()

    def get_noise_std(self):
        return self.cfg.algo.noise.std_max if self.noise_scheduler is None else self.noise_scheduler.val()

    def update_noise(self):
        if self.noise_scheduler is not None:
            self.noise_scheduler.step()

    def get_actions(self, obs, sample=True):
        """Policy action on rollout-normalised observations (no +-5 clamp on this side, torch_util.py:83-85), plus
        exploration noise: 'mixed' = per-env sigma spread over [std_min, std_max] along the global env axis."""
        x = self.obs_rms.normalize(obs) if self.cfg.algo.obs_norm else obs
        act = self.actor(x)
        if not sample:
            return act
        noise = self.cfg.algo.noise
        if noise.type == "mixed":
            return add_mixed_normal_noise(act, std_min=noise.std_min, std_max=noise.std_max, out_bounds=[-1., 1.],
                                          env_offset=self.env_offset, total_envs=self.total_envs)
        if noise.type == "fixed":
            return add_normal_noise(act, std=self.get_noise_std(), out_bounds=[-1., 1.])
        raise NotImplementedError(noise.type)

    # ---- rollout --------------------------------------------------------------------------------
    def _trajectory_slabs(self, T):
        sl = self._slabs.get(T)
        if sl is None:
            n, dev = self.cfg.num_envs, self.sim_device
            O = self.obs_dim[0] if not isinstance(self.obs_dim, int) else self.obs_dim
            mk = lambda *shape: torch.empty(shape, device=dev)  # noqa: E731
            sl = dict(obs=mk(n, T, O), act=mk(n, T, self.action_dim), rew=mk(n, T, 1), nobs=mk(n, T, O), done=mk(n, T, 1))
            self._slabs[T] = sl
        return sl

    @torch.no_grad()
    def explore_env(self, env, timesteps: int, random: bool):
        """Step the vectorised env `timesteps` times and return `(obs for the P-learner, 5-tuple for the V-learner,
        env steps taken)` -- the contract of pql_actor.py:87-127.  Everything stays on the GPU and nothing synchronises
        with the host: running statistics, trackers, the n-step window and the hand-off copies are all stream work."""
        algo, n = self.cfg.algo, self.cfg.num_envs
        sl = self._trajectory_slabs(timesteps)
        obs = self.obs
        for t in range(timesteps):
            if self.obs_rms is not None:
                self.obs_rms.update(obs)
            if random:   # warm-up: U(-1, 1) actions
                action = torch.rand((n, self.action_dim), device=self.sim_device).mul_(2.0).sub_(1.0)
            else:
                action = self.get_actions(obs, sample=True)
            next_obs, reward, done, info = env.step(action)
            self.update_tracker(reward, done, info)
            if algo.handle_timeout:
                done = handle_timeout(done, info)
            sl["obs"][:, t] = obs
            sl["act"][:, t] = action
            sl["rew"][:, t, 0] = reward
            sl["nobs"][:, t] = next_obs
            sl["done"][:, t, 0] = done
            obs = next_obs
        self.obs = obs
        rew = sl["rew"] * algo.reward_scale
        out = self.n_step_buffer.add_to_buffer(sl["obs"], sl["act"], rew, sl["nobs"], sl["done"])
        O = out[0].shape[-1]
        p_data = out[0].reshape(-1, O).to(self.p_learner_device, non_blocking=True)
        v_data = tuple(x.to(self.v_learner_device, non_blocking=True) for x in out)
        return p_data, v_data, timesteps * n

    def update_tracker(self, reward, done, info):
        """Episode return / length windows, updated with masked scatters on the device (the reference's
        `torch.where(done)[0]` + `.tolist()` stalls the stream every env step)."""
        finished = done.bool()
        self.current_returns += reward
        self.current_lengths += 1
        self.return_tracker.update(self.current_returns, finished)
        self.step_tracker.update(self.current_lengths, finished)
        self.current_returns.masked_fill_(finished, 0)
        self.current_lengths.masked_fill_(finished, 0)
        return done

    def add_info_tracker_log(self, log_info):
        return log_info
