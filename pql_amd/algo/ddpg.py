"""Synchronous DDPG on the same kernels (BASELINE config #1 plumbing).

Mirrors the hot part of `pql/algo/ddpg.py`: `AgentDDPG.update_net(memory)` = `update_times` x {sample,
obs_rms.normalize (NO clamp, ddpg.py:124-126), critic step (:147-157), actor step (:159-166), Polyak of the critic target and,
with `no_tgt_actor=False`, of the target actor (:134-135; `no_tgt_actor=True`, the shipped default: the target actor IS the actor, :22).  It is the V- and P-learner launch sequences
run back to back on one shared replay sample; the ActorCriticBase plumbing of the fork (bidex, success
trackers) is out of scope.
"""
from __future__ import annotations

import ctypes as C
from copy import deepcopy

import numpy as np
import torch

from pql_amd import _lib as L
from pql_amd.algo.pql_actor import PQLActor
from pql_amd.algo.pql_v_learner import LOSS_RING, _AdamState, _cfg_get, apply_optimizer
from pql_amd.models import model_name_to_path
from pql_amd.models.mlp import default_splits, mlp_forward_raw, output_view
from pql_amd.utils.common import load_class_from_path


class AgentDDPG(PQLActor):
    def __init__(self, env, cfg):
        cfg.algo.v_learner_gpu = cfg.algo.get("v_learner_gpu", 0) or 0
        cfg.algo.p_learner_gpu = cfg.algo.get("p_learner_gpu", 0) or 0
        super().__init__(env, cfg)
        self.device = self.sim_device
        algo = cfg.algo
        hidden = _cfg_get(algo, "hidden_layers")
        hidden = list(hidden) if hidden is not None else None
        act_class = load_class_from_path(algo.act_class, model_name_to_path[algo.act_class])
        cri_class = load_class_from_path(algo.cri_class, model_name_to_path[algo.cri_class])
        with torch.cuda.device(self.device):
            self.actor = act_class(self.obs_dim, self.action_dim, hidden_layers=hidden).to(self.device)
            self.critic = cri_class(self.obs_dim, self.action_dim, hidden_layers=hidden).to(self.device)
        self.critic_target = deepcopy(self.critic)
        # ddpg.py:21-22: a Polyak-averaged copy of the actor, or (no_tgt_actor=True, every shipped config) the actor itself
        self.actor_target = self.actor if algo.no_tgt_actor else deepcopy(self.actor)
        self.aopt, self.copt = _AdamState(self.actor.arena.data), _AdamState(self.critic.arena.data)
        self.closs = torch.zeros(LOSS_RING, device=self.device)
        self.aloss = torch.zeros(LOSS_RING, device=self.device)
        self._ws = None

    def explore_env(self, env, timesteps, random=False):
        n0 = self.n_step_buffer.nstep_count
        act_data, cri_data, steps = super().explore_env(env, timesteps, random)
        del act_data, n0
        return cri_data, steps

    def _workspace(self, B):
        if self._ws is not None and self._ws["B"] == B:
            return self._ws
        f = dict(dtype=torch.float32, device=self.device)
        O, A = self.obs_dim[0], self.action_dim
        al, cl = self.actor.layout, self.critic.layout
        ws = dict(B=B, ld_sa=L.ld(O + A), ld_o=L.ld(O), ld_a=L.ld(A), splits=default_splits(B))
        for k, shape in dict(x_sa=(B, ws["ld_sa"]), xn_sa=(B, ws["ld_sa"]), xn_obs=(B, ws["ld_o"]), x_obs=(B, ws["ld_o"]),
                             x_pi=(B, ws["ld_sa"]), rew=(B,), done=(B,), draw=(B, A), dy=(2, B, cl.ld_out),
                             dz_a=(1, B, ws["ld_a"]), gc=(cl.total,), ga=(al.total,), scratch=(2048,)).items():
            ws[k] = torch.zeros(shape, **f)
        ws["acts_a"] = torch.empty(al.acts_floats(B), **f)
        ws["acts_t"] = torch.empty(cl.acts_floats(B), **f)
        ws["acts_c"] = torch.empty(cl.acts_floats(B), **f)
        ws["bwd_c"] = torch.empty(cl.bwd_ws_floats(B, ws["splits"]), **f)
        ws["bwd_a"] = torch.empty(al.bwd_ws_floats(B, ws["splits"]), **f)
        self._ws = ws
        return ws

    @torch.no_grad()
    def update_once(self, memory, indices=None, noise=None):
        """One inner iteration of update_net; returns nothing (losses land in device rings)."""
        algo, dev = self.cfg.algo, self.device
        B = int(algo.batch_size)
        ws = self._workspace(B)
        O, A = self.obs_dim[0], self.action_dim
        al, cl = self.actor.layout, self.critic.layout
        with torch.cuda.device(dev):
            st = L.stream(dev)
            idx = memory.draw_indices(B) if indices is None else indices.to(dev, torch.int64).contiguous()
            draw = ws["draw"].normal_() if noise is None else noise.to(dev, torch.float32).contiguous()
            mean = var = None
            eps = 0.0
            if algo.obs_norm:
                mean, var, eps = self.obs_rms.get_states()
                mean, var = mean.contiguous(), var.contiguous()
            # sample + normalise WITHOUT clamp; x_obs (= norm(obs)) doubles as the actor-step input
            L.check(L.lib.pqlk_replay_gather_fused(C.byref(memory.ring.desc), L.ptr(idx), B, L.ptr(mean), L.ptr(var), float(eps), 0,
                                                   L.ptr(ws["x_sa"]), ws["ld_sa"], L.ptr(ws["xn_sa"]), L.ptr(ws["xn_obs"]), ws["ld_o"],
                                                   L.ptr(ws["rew"]), L.ptr(ws["done"]), st))
            ws["x_obs"][:, :O].copy_(ws["x_sa"][:, :O])
            ws["x_pi"][:, :O].copy_(ws["x_sa"][:, :O])
            # ---- critic step (ddpg.py:147-157)
            mlp_forward_raw(al, self.actor_target.arena.data, ws["xn_obs"], L.ACT_TANH_NOISE, draw, algo.noise.tgt_pol_std,
                            algo.noise.tgt_pol_noise_bound, ws["acts_a"], ws["xn_sa"][:, O:])
            mlp_forward_raw(cl, self.critic_target.arena.data, ws["xn_sa"], L.ACT_NONE, acts=ws["acts_t"])
            mlp_forward_raw(cl, self.critic.arena.data, ws["x_sa"], L.ACT_NONE, acts=ws["acts_c"])
            q, qt = output_view(cl, ws["acts_c"], B), output_view(cl, ws["acts_t"], B)
            L.check(L.lib.pqlk_td_mse_loss(L.ptr(q), L.ptr(qt), cl.ld_out, L.ptr(ws["rew"]), L.ptr(ws["done"]),
                                           float(algo.gamma) ** int(algo.nstep), B, L.ptr(ws["dy"]), L.ptr(self.closs),
                                           L.ptr(self.copt.step), LOSS_RING, L.ptr(ws["scratch"]), st))
            L.check(L.lib.pqlk_mlp_backward(C.byref(cl.desc), L.ptr(self.critic.arena.data), L.ptr(ws["x_sa"]), ws["ld_sa"], B,
                                            L.ptr(ws["acts_c"]), L.ptr(ws["dy"]), L.ptr(ws["gc"]), ws["splits"], None, 0, 0, 0, None, 0,
                                            L.ptr(ws["bwd_c"]), ws["bwd_c"].numel(), st))
            apply_optimizer(self.critic.arena.data, ws["gc"], self.copt, None, algo.critic_lr, algo.max_grad_norm, 0.0, 1.0, dev)
            # ---- actor step through the UPDATED critic (ddpg.py:159-166)
            mlp_forward_raw(al, self.actor.arena.data, ws["x_obs"], L.ACT_TANH, acts=ws["acts_a"], out2=ws["x_pi"][:, O:])
            mlp_forward_raw(cl, self.critic.arena.data, ws["x_pi"], L.ACT_NONE, acts=ws["acts_c"])
            L.check(L.lib.pqlk_dpg_loss(L.ptr(output_view(cl, ws["acts_c"], B)), cl.ld_out, 1, None, B, L.ptr(ws["dy"]),
                                        L.ptr(self.aloss), L.ptr(self.aopt.step), LOSS_RING, L.ptr(ws["scratch"]), st))
            a_out = output_view(al, ws["acts_a"], B)
            L.check(L.lib.pqlk_mlp_backward(C.byref(cl.desc), L.ptr(self.critic.arena.data), L.ptr(ws["x_pi"]), ws["ld_sa"], B,
                                            L.ptr(ws["acts_c"]), L.ptr(ws["dy"]), None, 1, L.ptr(ws["dz_a"]), ws["ld_a"], O, A,
                                            L.ptr(a_out), ws["ld_a"], L.ptr(ws["bwd_c"]), ws["bwd_c"].numel(), st))
            L.check(L.lib.pqlk_mlp_backward(C.byref(al.desc), L.ptr(self.actor.arena.data), L.ptr(ws["x_obs"]), ws["ld_o"], B,
                                            L.ptr(ws["acts_a"]), L.ptr(ws["dz_a"]), L.ptr(ws["ga"]), ws["splits"], None, 0, 0, 0, None, 0,
                                            L.ptr(ws["bwd_a"]), ws["bwd_a"].numel(), st))
            apply_optimizer(self.actor.arena.data, ws["ga"], self.aopt, None, algo.actor_lr, algo.max_grad_norm, 0.0, 1.0, dev)
            # ---- soft_update(critic_target, critic, tau)
            L.check(L.lib.pqlk_polyak(L.ptr(self.critic_target.arena.data), L.ptr(self.critic.arena.data),
                                      self.critic.arena.numel(), float(algo.tau), st))
            if self.actor_target is not self.actor:   # ddpg.py:134-135
                L.check(L.lib.pqlk_polyak(L.ptr(self.actor_target.arena.data), L.ptr(self.actor.arena.data),
                                          self.actor.arena.numel(), float(algo.tau), st))

    def update_net(self, memory):
        n = int(self.cfg.algo.update_times)
        for _ in range(n):
            self.update_once(memory)
        c, a = self.closs.tolist(), self.aloss.tolist()
        k = min(n, LOSS_RING)
        return {"train/critic_loss": float(np.mean(c[:k])), "train/actor_loss": float(np.mean(a[:k])),
                "train/return": self.return_tracker.mean(), "train/episode_length": self.step_tracker.mean()}
