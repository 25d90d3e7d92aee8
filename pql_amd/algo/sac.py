"""Synchronous SAC on the PQL kernels.

Mirrors the hot part of `pql/algo/sac.py`: `AgentSAC.update_net(memory)` = `update_times` x {sample, `obs_rms.normalize`
(no clamp), `update_critic` (:138-146: entropy-regularised n-step target through the squashed-Gaussian policy, twin MSE),
`update_actor` (:148-161: mean(alpha log pi - min Q) through the UPDATED critic, then the temperature step on
`log_alpha`), Polyak on the critic (and on the otherwise unused target policy when `no_tgt_actor=False`)}.  Everything the DDPG baseline already runs (gather,
fp32-MFMA MLP forward/backward, TD loss, clip+AdamW, Polyak) is reused; the new math is four small launches
(`pql_amd/csrc/sac.hip`): the policy head forward / backward and the two temperature kernels.  `log_alpha` lives on the
device and is read there by the kernels that need alpha, so an update has no host round trip and is graph-capturable.
RNG order per update = the reference's: replay indices, the rsample draw of the critic step, the rsample draw of the
actor step.
"""
from __future__ import annotations

import ctypes as C
import math
from copy import deepcopy

import numpy as np
import torch

from pql_amd import _lib as L
from pql_amd.algo.pql_actor import PQLActor
from pql_amd.algo.pql_v_learner import LOSS_RING, _AdamState, _cfg_get, apply_optimizer
from pql_amd.models import model_name_to_path
from pql_amd.models.mlp import default_splits, mlp_forward_raw, output_view
from pql_amd.utils.common import load_class_from_path


class AgentSAC(PQLActor):
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
        if self.actor.layout.dims[-1] != 2 * self.action_dim:
            raise ValueError("SAC needs a policy with a [mu | log_std] head (act_class: TanhDiagGaussianMLPPolicy)")
        self.critic_target = deepcopy(self.critic)
        # sac.py:19,104-105: with no_tgt_actor=False the reference keeps a Polyak-averaged copy of the policy; its critic target
        # (sac.py:138-146) samples the next action from `self.actor` either way, so the copy is state only (checkpoints)
        self.actor_target = self.actor if algo.no_tgt_actor else deepcopy(self.actor)
        self.aopt, self.copt = _AdamState(self.actor.arena.data), _AdamState(self.critic.arena.data)
        # temperature (sac.py:22-26,32-42): learned log_alpha starting at 0, or a fixed alpha from the config
        self.learn_alpha = algo.alpha is None
        self.log_alpha = torch.full((1,), 0.0 if self.learn_alpha else math.log(float(algo.alpha)), dtype=torch.float32, device=self.device)
        self.alpha_opt = _AdamState(self.log_alpha)
        self.target_entropy = -float(self.action_dim)
        self.closs = torch.zeros(LOSS_RING, device=self.device)
        self.aloss = torch.zeros(LOSS_RING, device=self.device)
        self.alpha_loss = torch.zeros(1, device=self.device)
        self._ws = None

    # ---- acting ------------------------------------------------------------------------------------
    def get_alpha(self, detach=True, scalar=False):
        alpha = self.log_alpha.exp()
        return float(alpha) if scalar else alpha

    def get_actions(self, obs, sample=True):
        x = self.obs_rms.normalize(obs) if self.cfg.algo.obs_norm else obs
        return self.actor.get_actions(x, sample=sample)

    def explore_env(self, env, timesteps, random=False):
        act_data, cri_data, steps = super().explore_env(env, timesteps, random)
        del act_data
        return cri_data, steps

    # ---- learning ----------------------------------------------------------------------------------
    def _workspace(self, B):
        if self._ws is not None and self._ws["B"] == B:
            return self._ws
        f = dict(dtype=torch.float32, device=self.device)
        O, A = self.obs_dim[0], self.action_dim
        al, cl = self.actor.layout, self.critic.layout
        ws = dict(B=B, ld_sa=L.ld(O + A), ld_o=L.ld(O), splits=default_splits(B))
        for k, shape in dict(x_sa=(B, ws["ld_sa"]), xn_sa=(B, ws["ld_sa"]), xn_obs=(B, ws["ld_o"]), x_obs=(B, ws["ld_o"]),
                             x_pi=(B, ws["ld_sa"]), rew=(B,), done=(B,), eps_next=(B, A), eps_cur=(B, A), logp_next=(B,), logp=(B,),
                             dy=(2, B, cl.ld_out), dx_pi=(B, ws["ld_sa"]), dy_a=(1, B, al.ld_out), gc=(cl.total,), ga=(al.total,),
                             g_alpha=(1,), scratch=(2048,)).items():
            ws[k] = torch.zeros(shape, **f)
        ws["acts_a"] = torch.empty(al.acts_floats(B), **f)
        ws["acts_t"] = torch.empty(cl.acts_floats(B), **f)
        ws["acts_c"] = torch.empty(cl.acts_floats(B), **f)
        ws["bwd_c"] = torch.empty(cl.bwd_ws_floats(B, ws["splits"]), **f)
        ws["bwd_a"] = torch.empty(al.bwd_ws_floats(B, ws["splits"]), **f)
        self._ws = ws
        return ws

    @torch.no_grad()
    def update_once(self, memory, indices=None, eps_next=None, eps_cur=None):
        """One inner iteration of update_net (losses land in device rings).  indices / eps_*: injected draws for parity tests."""
        algo, dev = self.cfg.algo, self.device
        B = int(algo.batch_size)
        ws = self._workspace(B)
        O, A = self.obs_dim[0], self.action_dim
        al, cl = self.actor.layout, self.critic.layout
        with torch.cuda.device(dev):
            st = L.stream(dev)
            idx = memory.draw_indices(B) if indices is None else indices.to(dev, torch.int64).contiguous()
            e_next = ws["eps_next"].normal_() if eps_next is None else eps_next.to(dev, torch.float32).contiguous()
            e_cur = ws["eps_cur"].normal_() if eps_cur is None else eps_cur.to(dev, torch.float32).contiguous()
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
            # ---- critic step (sac.py:138-146): y = r + (1-d) gamma^n (min Q_t(s', a') - alpha log pi(a'|s')), a' ~ pi(.|s')
            mlp_forward_raw(al, self.actor.arena.data, ws["xn_obs"], L.ACT_NONE, acts=ws["acts_a"])
            y_a = output_view(al, ws["acts_a"], B)[0]
            L.check(L.lib.pqlk_sg_head_forward(L.ptr(y_a), al.ld_out, L.ptr(e_next), B, A, L.ptr(ws["xn_sa"][:, O:]), ws["ld_sa"],
                                               L.ptr(ws["logp_next"]), st))
            mlp_forward_raw(cl, self.critic_target.arena.data, ws["xn_sa"], L.ACT_NONE, acts=ws["acts_t"])
            mlp_forward_raw(cl, self.critic.arena.data, ws["x_sa"], L.ACT_NONE, acts=ws["acts_c"])
            q, qt = output_view(cl, ws["acts_c"], B), output_view(cl, ws["acts_t"], B)
            L.check(L.lib.pqlk_sac_entropy_shift(L.ptr(qt), cl.ld_out, B * cl.ld_out, 2, L.ptr(ws["logp_next"]), L.ptr(self.log_alpha), B, st))
            L.check(L.lib.pqlk_td_mse_loss(L.ptr(q), L.ptr(qt), cl.ld_out, L.ptr(ws["rew"]), L.ptr(ws["done"]),
                                           float(algo.gamma) ** int(algo.nstep), B, L.ptr(ws["dy"]), L.ptr(self.closs),
                                           L.ptr(self.copt.step), LOSS_RING, L.ptr(ws["scratch"]), st))
            L.check(L.lib.pqlk_mlp_backward(C.byref(cl.desc), L.ptr(self.critic.arena.data), L.ptr(ws["x_sa"]), ws["ld_sa"], B,
                                            L.ptr(ws["acts_c"]), L.ptr(ws["dy"]), L.ptr(ws["gc"]), ws["splits"], None, 0, 0, 0, None, 0,
                                            L.ptr(ws["bwd_c"]), ws["bwd_c"].numel(), st))
            apply_optimizer(self.critic.arena.data, ws["gc"], self.copt, None, algo.critic_lr, algo.max_grad_norm, 0.0, 1.0, dev)
            # ---- actor step through the UPDATED critic (sac.py:148-153): L = mean(alpha log pi(a|s) - min Q(s, a)), a ~ pi(.|s)
            mlp_forward_raw(al, self.actor.arena.data, ws["x_obs"], L.ACT_NONE, acts=ws["acts_a"])
            L.check(L.lib.pqlk_sg_head_forward(L.ptr(y_a), al.ld_out, L.ptr(e_cur), B, A, L.ptr(ws["x_pi"][:, O:]), ws["ld_sa"],
                                               L.ptr(ws["logp"]), st))
            mlp_forward_raw(cl, self.critic.arena.data, ws["x_pi"], L.ACT_NONE, acts=ws["acts_c"])
            L.check(L.lib.pqlk_dpg_loss(L.ptr(output_view(cl, ws["acts_c"], B)), cl.ld_out, 1, None, B, L.ptr(ws["dy"]),
                                        L.ptr(self.aloss), L.ptr(self.aopt.step), LOSS_RING, L.ptr(ws["scratch"]), st))
            # temperature terms with the alpha the actor loss uses (before its own update): actor loss += alpha mean(log pi),
            # g_alpha = alpha * mean(-log pi - target_entropy)
            L.check(L.lib.pqlk_sac_alpha_terms(L.ptr(ws["logp"]), B, L.ptr(self.log_alpha), self.target_entropy, L.ptr(ws["g_alpha"]),
                                               L.ptr(self.alpha_loss), L.ptr(self.aloss), L.ptr(self.aopt.step), LOSS_RING, st))
            # dL/da = -(1/B) d minQ / da : the critic's input gradient, action columns [O, O+A) of dx_pi
            L.check(L.lib.pqlk_mlp_backward(C.byref(cl.desc), L.ptr(self.critic.arena.data), L.ptr(ws["x_pi"]), ws["ld_sa"], B,
                                            L.ptr(ws["acts_c"]), L.ptr(ws["dy"]), None, 1, L.ptr(ws["dx_pi"]), ws["ld_sa"], 0, 0,
                                            None, 0, L.ptr(ws["bwd_c"]), ws["bwd_c"].numel(), st))
            L.check(L.lib.pqlk_sg_head_backward(L.ptr(y_a), al.ld_out, L.ptr(e_cur), L.ptr(ws["x_pi"][:, O:]), ws["ld_sa"],
                                                L.ptr(ws["dx_pi"][:, O:]), ws["ld_sa"], L.ptr(self.log_alpha), 1.0 / B, B, A,
                                                L.ptr(ws["dy_a"]), st))
            L.check(L.lib.pqlk_mlp_backward(C.byref(al.desc), L.ptr(self.actor.arena.data), L.ptr(ws["x_obs"]), ws["ld_o"], B,
                                            L.ptr(ws["acts_a"]), L.ptr(ws["dy_a"]), L.ptr(ws["ga"]), ws["splits"], None, 0, 0, 0, None, 0,
                                            L.ptr(ws["bwd_a"]), ws["bwd_a"].numel(), st))
            apply_optimizer(self.actor.arena.data, ws["ga"], self.aopt, None, algo.actor_lr, algo.max_grad_norm, 0.0, 1.0, dev)
            if self.learn_alpha:   # sac.py:155-157: AdamW (torch defaults, incl. weight decay) on the single scalar, clipped like the rest
                apply_optimizer(self.log_alpha, ws["g_alpha"], self.alpha_opt, None, algo.alpha_lr, algo.max_grad_norm, 0.0, 1.0, dev)
            # ---- soft_update(critic_target, critic, tau)
            L.check(L.lib.pqlk_polyak(L.ptr(self.critic_target.arena.data), L.ptr(self.critic.arena.data),
                                      self.critic.arena.numel(), float(algo.tau), st))
            if self.actor_target is not self.actor:   # sac.py:104-105
                L.check(L.lib.pqlk_polyak(L.ptr(self.actor_target.arena.data), L.ptr(self.actor.arena.data),
                                          self.actor.arena.numel(), float(algo.tau), st))

    def update_net(self, memory):
        n = int(self.cfg.algo.update_times)
        for _ in range(n):
            self.update_once(memory)
        c, a = self.closs.tolist(), self.aloss.tolist()
        k = min(n, LOSS_RING)
        return {"train/critic_loss": float(np.mean(c[:k])), "train/actor_loss": float(np.mean(a[:k])),
                "train/return": self.return_tracker.mean(), "train/episode_length": self.step_tracker.mean(),
                "train/alpha": self.get_alpha(scalar=True)}
