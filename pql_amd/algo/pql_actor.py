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
        self.env = env
        self.cfg = cfg
        self.obs_dim = self.env.observation_space.shape
        self.action_dim = self.env.action_space.shape[0]
        self.sim_device = torch.device(f"{cfg.sim_device}")
        self.v_learner_device = torch.device(f"cuda:{cfg.algo.v_learner_gpu}")
        self.p_learner_device = torch.device(f"cuda:{cfg.algo.p_learner_gpu}")
        self.env_offset, self.total_envs = int(env_offset), total_envs
        self.actor = None
        self.obs = None
        n = cfg.num_envs
        self.return_tracker = DeviceTracker(cfg.algo.tracker_len, self.sim_device)
        self.step_tracker = DeviceTracker(cfg.algo.tracker_len, self.sim_device)
        self.current_returns = torch.zeros(n, dtype=torch.float32, device=self.sim_device)
        self.current_lengths = torch.zeros(n, dtype=torch.float32, device=self.sim_device)
        if cfg.info_track_keys is not None:
            raise NotImplementedError("info_track_keys needs a simulator's info dict; out of scope")
        self.obs_rms = RunningMeanStd(shape=self.obs_dim, device=self.sim_device) if cfg.algo.obs_norm else None
        self.n_step_buffer = NStepReplay(self.obs_dim, self.action_dim, n, cfg.algo.nstep, device=self.sim_device)
        noise = cfg.algo.noise
        if noise.decay == "linear":
            self.noise_scheduler = LinearSchedule(noise.std_max, noise.std_min, noise.lin_decay_iters)
        elif noise.decay == "exp":
            self.noise_scheduler = ExponentialSchedule(noise.std_max, noise.exp_decay_rate, noise.std_min)
        else:
            self.noise_scheduler = None

    def reset_agent(self):
        self.obs = self.env.reset()

    def get_noise_std(self):
        return self.cfg.algo.noise.std_max if self.noise_scheduler is None else self.noise_scheduler.val()

    def update_noise(self):
        if self.noise_scheduler is not None:
            self.noise_scheduler.step()

    def get_actions(self, obs, sample=True):
        if self.cfg.algo.obs_norm:
            obs = self.obs_rms.normalize(obs)      # rollout side: no clamp (torch_util.py:83-85)
        actions = self.actor(obs)
        if sample:
            noise = self.cfg.algo.noise
            if noise.type == "fixed":
                actions = add_normal_noise(actions, std=self.get_noise_std(), out_bounds=[-1., 1.])
            elif noise.type == "mixed":
                actions = add_mixed_normal_noise(actions, std_min=noise.std_min, std_max=noise.std_max, out_bounds=[-1., 1.],
                                                 env_offset=self.env_offset, total_envs=self.total_envs)
            else:
                raise NotImplementedError(noise.type)
        return actions

    @torch.no_grad()
    def explore_env(self, env, timesteps: int, random: bool):
        n, dev = self.cfg.num_envs, self.sim_device
        O = self.obs_dim[0] if not isinstance(self.obs_dim, int) else self.obs_dim
        traj_states = torch.empty((n, timesteps, O), device=dev)
        traj_actions = torch.empty((n, timesteps, self.action_dim), device=dev)
        traj_rewards = torch.empty((n, timesteps), device=dev)
        traj_next_states = torch.empty((n, timesteps, O), device=dev)
        traj_dones = torch.empty((n, timesteps), device=dev)
        obs = self.obs
        for i in range(timesteps):
            if self.cfg.algo.obs_norm:
                self.obs_rms.update(obs)
            if random:
                action = torch.rand((n, self.action_dim), device=dev) * 2.0 - 1.0
            else:
                action = self.get_actions(obs, sample=True)
            next_obs, reward, done, info = env.step(action)
            self.update_tracker(reward, done, info)
            if self.cfg.algo.handle_timeout:
                done = handle_timeout(done, info)
            traj_states[:, i] = obs
            traj_actions[:, i] = action
            traj_dones[:, i] = done
            traj_rewards[:, i] = reward
            traj_next_states[:, i] = next_obs
            obs = next_obs
        self.obs = obs
        traj_rewards = self.cfg.algo.reward_scale * traj_rewards.reshape(n, timesteps, 1)
        traj_dones = traj_dones.reshape(n, timesteps, 1)
        obs, action, reward, next_obs, done = self.n_step_buffer.add_to_buffer(traj_states, traj_actions, traj_rewards,
                                                                               traj_next_states, traj_dones)
        act_data = obs.reshape(-1, O).to(self.p_learner_device, non_blocking=True)
        cri_data = tuple(t.to(self.v_learner_device, non_blocking=True) for t in (obs, action, reward, next_obs, done))
        return act_data, cri_data, timesteps * n

    def update_tracker(self, reward, done, info):
        self.current_returns += reward
        self.current_lengths += 1
        d = done.bool()
        self.return_tracker.update(self.current_returns, d)
        self.step_tracker.update(self.current_lengths, d)
        self.current_returns.masked_fill_(d, 0)
        self.current_lengths.masked_fill_(d, 0)
        return done

    def add_info_tracker_log(self, log_info):
        return log_info
