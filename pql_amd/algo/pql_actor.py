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
from pql_amd.utils import handoff as H
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
        self.ptr = torch.zeros(1, dtype=torch.int64, device=device)   # write pointer, updated in place (the HIP rollout step shares it)

    def update(self, values, mask):
        """deque.extend(values[mask]) of common.Tracker: when more than max_len episodes finish in one step only the LAST
        max_len of them (in env order) stay, so every kept value has a slot of its own and the scatter is deterministic."""
        rank = torch.cumsum(mask.to(torch.int64), 0)
        total = rank[-1]
        keep = mask & (rank > total - self.max_len)
        pos = (self.ptr + rank - 1) % self.max_len
        self.ring.scatter_(0, torch.where(keep, pos, torch.full_like(pos, self.max_len)), values)
        self.ptr.copy_((self.ptr + total) % self.max_len)

    def mean(self):
        return float(self.ring[: self.max_len].mean())


class PQLActor:
    def __init__(self, env, cfg, env_offset=0, total_envs=None):
        self.env, self.cfg = env, cfg
        self.obs_dim = env.observation_space.shape
        self.action_dim = env.action_space.shape[0]
        self.sim_device = torch.device(f"{cfg.sim_device}")
        if self.sim_device.type == "cuda" and self.sim_device.index is None:
            self.sim_device = torch.device("cuda", torch.cuda.current_device())
        self.v_learner_device = torch.device(f"cuda:{cfg.algo.v_learner_gpu}")
        self.p_learner_device = torch.device(f"cuda:{cfg.algo.p_learner_gpu}")
        self.env_offset, self.total_envs = int(env_offset), total_envs   # position on the GLOBAL env axis (data parallel)
        self._actor = None  # rollout replica of the policy, assigned by the driver (`actor` is a property: see below)
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
        # own device generator for the exploration draws, as the reference's actor process has its own: the default generator
        # is switched into capture mode whenever ANY hipGraph is being captured (a learner thread re-capturing its step),
        # and an eager draw from it at that moment raises
        self.gen = None
        if self.sim_device.type == "cuda":
            self.gen = torch.Generator(device=self.sim_device)
            self.gen.manual_seed(int(torch.randint(0, 2 ** 62, (1,)).item()))
        self._pk = None        # fragment-ordered copy of the rollout replica's weights (fused policy forward), see set_actor
        self._pk_stale = True  # re-derived at the start of every explore_env and after set_actor / assignment of `.actor`
        self._fwd_buf = None   # (zero-padded input tile, activation scratch) of the fused policy forward
        self._slabs = {}   # (N, T, .) trajectory slabs, allocated once per horizon length and reused
        # n-step output blocks handed to the learners: OUT_BLOCKS per block size, each with a lease so that the rollout
        # stream re-uses one only after the learners' streams have inserted it (pql_amd.utils.handoff)
        self._out_blocks, self._out_next = {}, {}

    OUT_BLOCKS = 3

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

    def get_actions(self, obs, sample=True, draw=None):
        """Policy action on rollout-normalised observations (no +-5 clamp on this side, torch_util.py:83-85), plus
        exploration noise: 'mixed' = per-env sigma spread over [std_min, std_max] along the global env axis."""
        act = self._policy_forward(obs)
        if not sample:
            return act
        noise = self.cfg.algo.noise
        if noise.type == "mixed":
            return add_mixed_normal_noise(act, std_min=noise.std_min, std_max=noise.std_max, out_bounds=[-1., 1.],
                                          env_offset=self.env_offset, total_envs=self.total_envs, generator=self.gen, draw=draw)
        if noise.type == "fixed":
            return add_normal_noise(act, std=self.get_noise_std(), out_bounds=[-1., 1.], generator=self.gen, draw=draw)
        raise NotImplementedError(noise.type)

    @property
    def actor(self):
        return self._actor

    @actor.setter
    def actor(self, module):   # the reference idiom `pql_actor.actor = deepcopy(actor).to(sim_device)` (train_pql.py:52,109)
        self._actor = module
        self._pk_stale = True

    def _policy_forward(self, obs):
        """tanh(MLP(normalise(obs))) of the rollout replica.  TanhMLPPolicy on the GPU: the observation is normalised straight
        into the policy's zero-padded input tile and the whole network is ONE fused launch writing a contiguous (N, A) action
        matrix (`pqlk_mlp_forward` with the fragment-ordered weight copy refreshed in set_actor); anything else goes through
        the module's own forward."""
        from pql_amd import _lib as L
        from pql_amd.models.mlp import mlp_forward_raw
        if self._pk_stale:
            self._refresh_packed()
        pk = self._pk
        if (pk is None or pk.tensor is None or getattr(self.actor, "out_act", None) != L.ACT_TANH or not obs.is_cuda
                or obs.dtype != torch.float32 or not obs.is_contiguous() or obs.dim() != 2):
            return self.actor(self.obs_rms.normalize(obs) if self.cfg.algo.obs_norm else obs)
        lay, n = self.actor.layout, obs.shape[0]
        buf = self._fwd_buf
        if buf is None or buf[0].shape[0] != n:
            dev = obs.device
            buf = self._fwd_buf = (torch.zeros((n, lay.ld_in), dtype=torch.float32, device=dev),
                                   torch.empty(lay.acts_floats(n), dtype=torch.float32, device=dev))
        x_pad, acts = buf
        if self.cfg.algo.obs_norm:
            self.obs_rms.normalize(obs, out=x_pad)
        else:
            x_pad[:, : obs.shape[1]].copy_(obs)
        act = torch.empty((n, self.action_dim), dtype=torch.float32, device=obs.device)
        mlp_forward_raw(lay, self.actor.arena.data, x_pad, L.ACT_TANH, acts=acts, out2=act, packed=pk, stash_all=False)
        return act

    def _refresh_packed(self):
        from pql_amd.models.mlp import PackedWeights
        fused = self.cfg.algo.get("fused", True) if hasattr(self.cfg.algo, "get") else getattr(self.cfg.algo, "fused", True)
        if self.sim_device.type != "cuda" or not hasattr(self.actor, "layout") or not fused:   # algo.fused=False: per-layer GEMMs everywhere
            self._pk, self._pk_stale = None, False
            return
        if self._pk is None or self._pk.layout.dims != self.actor.layout.dims:
            self._pk = PackedWeights(self.actor.layout, self.sim_device)
            self._fwd_buf = None
        with torch.cuda.device(self.sim_device):
            self._pk.refresh(self.actor.arena.data)
        self._pk_stale = False

    @torch.no_grad()
    def set_actor(self, actor):
        """Adopt new policy weights into the rollout replica (train_pql.py:52,109 `pql_actor.actor = deepcopy(actor).to(
        sim_device)`): a fenced arena-to-arena copy; from another GPU it goes through the copy streams."""
        if self.actor is None or self.actor.layout.dims != actor.layout.dims:
            from copy import deepcopy
            self.actor = deepcopy(actor).to(self.sim_device)
            return
        if actor is self.actor:
            self._pk_stale = True   # the caller may have stepped or loaded it in place
            return
        st = torch.cuda.current_stream(self.sim_device)
        with H.LOCK:
            if H.crosses(actor.arena.device, self.sim_device):
                blk = H.shipper(actor.arena.device, self.sim_device, "params").ship((actor.arena.data,), H.lease_of(actor))
                lease = H.acquire(blk, st)
                self.actor.arena.data.copy_(blk[0], non_blocking=True)
            else:
                lease = H.acquire(actor, st)
                self.actor.arena.data.copy_(actor.arena.data, non_blocking=True)
            H.release(lease, st)
        self._pk_stale = True

    def _out_block(self, M):
        """Next n-step output block of M rows (round robin), reclaimed from its previous readers."""
        blocks = self._out_blocks.setdefault(M, [])
        k = self._out_next.get(M, 0)
        self._out_next[M] = (k + 1) % self.OUT_BLOCKS
        if k >= len(blocks):
            dev = self.sim_device
            O = self.obs_dim[0] if not isinstance(self.obs_dim, int) else self.obs_dim
            mk = lambda c: torch.empty((M, c), dtype=torch.float32, device=dev)  # noqa: E731
            blocks.append(H.Block((mk(O), mk(self.action_dim), mk(1), mk(O), mk(1)), H.Lease()))
        blk = blocks[k]
        with H.LOCK:
            H.reclaim(H.lease_of(blk), torch.cuda.current_stream(self.sim_device))
        return blk

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
    def explore_env(self, env, timesteps: int, random: bool, draws=None):
        """Step the vectorised env `timesteps` times and return `(obs for the P-learner, 5-tuple for the V-learner,
        env steps taken)` -- the contract of pql_actor.py:87-127.  Everything stays on the GPU and nothing synchronises
        with the host: running statistics, trackers, the n-step window and the hand-off copies are all stream work.
        `draws`: optional per-step (N, A) samples to use instead of this actor's generator (parity tests): U(0,1) when
        `random`, N(0,1) otherwise -- the two draws the reference makes (pql_actor.py:101, noise.py:34-35)."""
        algo, n = self.cfg.algo, self.cfg.num_envs
        sl = self._trajectory_slabs(timesteps)
        self._pk_stale = True   # the replica may have been stepped or loaded in place since the last call: re-pack once per call
        obs = self.obs
        for t in range(timesteps):
            if self.obs_rms is not None:
                self.obs_rms.update(obs)
            if random:   # warm-up: U(-1, 1) actions
                u = (torch.rand((n, self.action_dim), device=self.sim_device, generator=self.gen) if draws is None
                     else draws[t].to(self.sim_device, torch.float32).clone())
                action = u.mul_(2.0).sub_(1.0)
            else:
                action = (self.get_actions(obs, sample=True) if draws is None
                          else self.get_actions(obs, sample=True, draw=draws[t].to(self.sim_device)))
            next_obs, reward, done, info = env.step(action)
            if not self._bookkeep_hip(sl, t, timesteps, obs, action, next_obs, reward, done, info):
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
        blk = self._out_block(self.n_step_buffer.rows_out(timesteps))
        self.n_step_buffer.add_to_buffer(sl["obs"], sl["act"], rew, sl["nobs"], sl["done"], out=tuple(blk))
        sim = torch.cuda.current_stream(self.sim_device)
        with H.LOCK:
            H.lease_of(blk).ready = H._event(sim)
        # hand-off (pql_actor.py:121-126: `.to(learner_device)`): same GPU -> the block itself; another GPU -> copy
        # streams + landing blocks on that GPU.  Either way the learner's stream, not the host, waits for the data.
        v_data = blk if not H.crosses(self.sim_device, self.v_learner_device) else \
            H.shipper(self.sim_device, self.v_learner_device, "transitions").ship(tuple(blk), H.lease_of(blk))
        if self.p_learner_device == self.v_learner_device:
            p_data = H.tag(v_data[0].view(v_data[0].shape), H.lease_of(v_data))
        elif not H.crosses(self.sim_device, self.p_learner_device):
            p_data = H.tag(blk[0].view(blk[0].shape), H.lease_of(blk))
        else:
            p_data = H.shipper(self.sim_device, self.p_learner_device, "obs").ship((blk[0],), H.lease_of(blk))
            p_data = H.tag(p_data[0], H.lease_of(p_data))
        return p_data, v_data, timesteps * n

    def _bookkeep_hip(self, sl, t, T, obs, action, next_obs, reward, done, info):
        """Slab writes + episode accumulators + moving windows + handle_timeout of one env step as ONE launch
        (`pqlk_rollout_step`); False = shapes / dtypes the kernel does not take (the torch ops below then do the same)."""
        import ctypes as C
        from pql_amd import _lib as L
        if self.sim_device.type != "cuda" or done.dtype != torch.bool or reward.dtype != torch.float32:
            return False
        trunc = info.get("TimeLimit.truncated", None) if (self.cfg.algo.handle_timeout and isinstance(info, dict)) else None
        if trunc is not None and trunc.dtype != torch.bool:
            return False
        tens = (obs, action, next_obs, reward)
        if any(x.dtype != torch.float32 or not x.is_contiguous() for x in tens) or not done.is_contiguous():
            return False
        n, O = obs.shape[0], obs.shape[1]
        with torch.cuda.device(self.sim_device):
            L.check(L.lib.pqlk_rollout_step(n, O, self.action_dim, T, t, L.ptr(obs), L.ptr(action), L.ptr(next_obs), L.ptr(reward),
                                            C.c_void_p(done.data_ptr()), C.c_void_p(trunc.data_ptr()) if trunc is not None else None,
                                            L.ptr(sl["obs"]), L.ptr(sl["act"]), L.ptr(sl["rew"]), L.ptr(sl["nobs"]), L.ptr(sl["done"]),
                                            L.ptr(self.current_returns), L.ptr(self.current_lengths), L.ptr(self.return_tracker.ring),
                                            L.ptr(self.step_tracker.ring), L.ptr(self.return_tracker.ptr), L.ptr(self.step_tracker.ptr),
                                            self.return_tracker.max_len, L.stream(self.sim_device)))
        return True

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
