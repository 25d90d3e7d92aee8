"""Synchronous CrossQ on the PQL kernels.

Mirrors `pql/algo/crossQ.py`: `AgentCrossQ.update_net(memory)` = `update_times` x {sample, `obs_rms.normalize` (no
clamp), `update_critic` (:144-157: ONE joint forward of [obs; next_obs] with [action; target-policy action] through the
BatchNorm critic in training mode -- batch statistics over all 2B rows -- current Q from the first half, the target from the
detached second half, twin MSE), `update_actor` (:159-166: DPG through the critic, which stays in training mode: its
batch statistics are those of the B actor rows and the running statistics move again)}.  There is no target critic and,
with `no_tgt_actor=True` (the default), no target actor either (`False`: a Polyak-averaged copy supplies the target-policy actions).  The critic is `pql_amd.models.batchnorm.DoubleQBatchNorm` (one-layer GEMM
calls + the BatchNorm/ELU kernels of pql_amd/csrc/bn.hip); gather, actor, losses and the optimiser are the launches the
DDPG baseline uses.  RNG order per update: replay indices, then the target-policy noise draw.
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


class AgentCrossQ(PQLActor):
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
        if not hasattr(self.critic, "backward_raw"):
            raise ValueError("CrossQ needs the BatchNorm critic (cri_class: DoubleQBatchNorm)")
        # crossQ.py:21,71,132-133: the target-policy actions come from a Polyak-averaged copy unless no_tgt_actor=True (the default)
        self.actor_target = self.actor if algo.no_tgt_actor else deepcopy(self.actor)
        self.aopt, self.copt = _AdamState(self.actor.arena.data), _AdamState(self.critic.arena.data)
        self.closs = torch.zeros(LOSS_RING, device=self.device)
        self.aloss = torch.zeros(LOSS_RING, device=self.device)
        self._ws = None

    def explore_env(self, env, timesteps, random=False):
        act_data, cri_data, steps = super().explore_env(env, timesteps, random)
        del act_data
        return cri_data, steps

    def _workspace(self, B):
        if self._ws is not None and self._ws["B"] == B:
            return self._ws
        f = dict(dtype=torch.float32, device=self.device)
        O, A = self.obs_dim[0], self.action_dim
        al = self.actor.layout
        ws = dict(B=B, ld_sa=L.ld(O + A), ld_o=L.ld(O), ld_a=L.ld(A), splits=default_splits(B))
        ws["x_all"] = torch.zeros((2 * B, ws["ld_sa"]), **f)          # [obs | action] rows, then [next_obs | target action] rows
        ws["x_sa"], ws["xn_sa"] = ws["x_all"][:B], ws["x_all"][B:]
        for k, shape in dict(xn_obs=(B, ws["ld_o"]), x_obs=(B, ws["ld_o"]), x_pi=(B, ws["ld_sa"]), rew=(B,), done=(B,), draw=(B, A),
                             q=(2, B, 32), qt=(2, B, 32), dy=(2, B, 32), dq_all=(2, 2 * B, 32), dz_a=(1, B, ws["ld_a"]),
                             gc=(self.critic.total,), ga=(al.total,), scratch=(2048,)).items():
            ws[k] = torch.zeros(shape, **f)
        ws["acts_a"] = torch.empty(al.acts_floats(B), **f)
        ws["bwd_a"] = torch.empty(al.bwd_ws_floats(B, ws["splits"]), **f)
        self._ws = ws
        return ws

    @torch.no_grad()
    def update_once(self, memory, indices=None, noise=None):
        """One inner iteration of update_net (losses land in device rings).  indices / noise: injected draws for parity tests."""
        algo, dev = self.cfg.algo, self.device
        B = int(algo.batch_size)
        ws = self._workspace(B)
        O, A = self.obs_dim[0], self.action_dim
        al = self.actor.layout
        with torch.cuda.device(dev):
            st = L.stream(dev)
            idx = memory.draw_indices(B) if indices is None else indices.to(dev, torch.int64).contiguous()
            draw = ws["draw"].normal_() if noise is None else noise.to(dev, torch.float32).contiguous()
            mean = var = None
            eps = 0.0
            if algo.obs_norm:
                mean, var, eps = self.obs_rms.get_states()
                mean, var = mean.contiguous(), var.contiguous()
            L.check(L.lib.pqlk_replay_gather_fused(C.byref(memory.ring.desc), L.ptr(idx), B, L.ptr(mean), L.ptr(var), float(eps), 0,
                                                   L.ptr(ws["x_sa"]), ws["ld_sa"], L.ptr(ws["xn_sa"]), L.ptr(ws["xn_obs"]), ws["ld_o"],
                                                   L.ptr(ws["rew"]), L.ptr(ws["done"]), st))
            ws["x_obs"][:, :O].copy_(ws["x_sa"][:, :O])
            ws["x_pi"][:, :O].copy_(ws["x_sa"][:, :O])
            # ---- critic step (crossQ.py:144-157)
            mlp_forward_raw(al, self.actor_target.arena.data, ws["xn_obs"], L.ACT_TANH_NOISE, draw, algo.noise.tgt_pol_std,
                            algo.noise.tgt_pol_noise_bound, ws["acts_a"], ws["xn_sa"][:, O:])
            q_all = self.critic.forward_raw(ws["x_all"], training=True)          # (2, 2B, 32): batch statistics over all 2B rows
            ws["q"].copy_(q_all[:, :B]); ws["qt"].copy_(q_all[:, B:])            # current / (detached) next halves
            L.check(L.lib.pqlk_td_mse_loss(L.ptr(ws["q"]), L.ptr(ws["qt"]), 32, L.ptr(ws["rew"]), L.ptr(ws["done"]),
                                           float(algo.gamma) ** int(algo.nstep), B, L.ptr(ws["dy"]), L.ptr(self.closs),
                                           L.ptr(self.copt.step), LOSS_RING, L.ptr(ws["scratch"]), st))
            ws["dq_all"][:, :B].copy_(ws["dy"])                                  # the next-state rows carry no loss gradient
            self.critic.backward_raw(ws["x_all"], ws["dq_all"], grads=ws["gc"])
            apply_optimizer(self.critic.arena.data, ws["gc"], self.copt, None, algo.critic_lr, algo.max_grad_norm, 0.0, 1.0, dev)
            # ---- actor step (crossQ.py:159-166): the critic is still in training mode there
            mlp_forward_raw(al, self.actor.arena.data, ws["x_obs"], L.ACT_TANH, acts=ws["acts_a"], out2=ws["x_pi"][:, O:])
            q_pi = self.critic.forward_raw(ws["x_pi"], training=True)
            L.check(L.lib.pqlk_dpg_loss(L.ptr(q_pi), 32, 1, None, B, L.ptr(ws["dy"]), L.ptr(self.aloss), L.ptr(self.aopt.step),
                                        LOSS_RING, L.ptr(ws["scratch"]), st))
            dx = self.critic.backward_raw(ws["x_pi"], ws["dy"], grads=None, need_dx=True)
            a_out = output_view(al, ws["acts_a"], B)[0]
            ws["dz_a"][0, :, :A] = dx[:, O:O + A] * (1.0 - a_out[:, :A] * a_out[:, :A])       # through the actor's tanh
            L.check(L.lib.pqlk_mlp_backward(C.byref(al.desc), L.ptr(self.actor.arena.data), L.ptr(ws["x_obs"]), ws["ld_o"], B,
                                            L.ptr(ws["acts_a"]), L.ptr(ws["dz_a"]), L.ptr(ws["ga"]), ws["splits"], None, 0, 0, 0, None, 0,
                                            L.ptr(ws["bwd_a"]), ws["bwd_a"].numel(), st))
            apply_optimizer(self.actor.arena.data, ws["ga"], self.aopt, None, algo.actor_lr, algo.max_grad_norm, 0.0, 1.0, dev)
            if self.actor_target is not self.actor:   # crossQ.py:132-133
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
