"""CPU oracle for the PQL learner hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

A from-scratch restatement (torch CPU fp32 / numpy) of the arithmetic of the
reference path named in SURVEY.md section 8(a).  It exists to CHECK the HIP kernels
(tests/, __graft_entry__.smoke()) and to be TIMED as the CPU baseline
(bench.py `cpu_baseline`, kind="port").  Nothing under pql_amd/ may import it.

Pinning: every function here is asserted against golden vectors produced by the
reference itself (tools/gen_golden.py -> tests/golden/*.npz) in
tests/test_oracle_golden.py.  Third-party arithmetic (sgemm, exp/tanh, AdamW) is
torch 2.10 CPU, the same library the reference calls; the reference ships no
tests of its own, so those fixtures are the only pin.

Reference citations are path:line under the upstream tree (pql/...).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

HIDDEN_DEFAULT = (512, 256, 128)  # pql/models/mlp.py:32-33


# =========================================================================== ring buffer
def ring_plan(next_p: int, if_full: bool, capacity: int, m: int):
    """Pointer law of ReplayBuffer.add_to_buffer (pql/replay/simple_replay.py:52-83) and of the
    P-learner's inline ring (pql/algo/pql_p_learner.py:72-83).

    Returns (copies, new_next_p, new_if_full, cur_capacity); copies is an ordered list of
    (dst_start, src_start, length) applied in sequence (the tail copy wins on overlap).
    """
    p = next_p + m
    copies = []
    if p > capacity:
        if_full = True
        head = capacity - next_p
        if head > 0:
            copies.append((next_p, 0, head))
        p -= capacity
        if p > capacity:
            raise ValueError("insert larger than the ring can absorb (reference raises a shape error here)")
        copies.append((0, m - p, p))  # LAST p rows go to the front  (:66 `obs[-p:]`)
    else:
        copies.append((next_p, 0, m))
    cur = capacity if if_full else p
    return copies, p, if_full, cur


class RingRef:
    """SoA replay ring, V-learner flavour (obs, action, reward, next_obs, done[bool])."""

    def __init__(self, capacity: int, obs_dim: int, act_dim: int):
        self.capacity = int(capacity)
        self.obs_dim, self.act_dim = int(obs_dim), int(act_dim)
        self.next_p, self.if_full, self.cur_capacity = 0, False, 0
        self.obs = torch.zeros((self.capacity, obs_dim))
        self.act = torch.zeros((self.capacity, act_dim))
        self.rew = torch.zeros((self.capacity, 1))
        self.nobs = torch.zeros((self.capacity, obs_dim))
        self.done = torch.zeros((self.capacity, 1), dtype=torch.bool)  # stored as bool (:51 `.bool()`)

    def insert(self, obs, act, rew, nobs, done):
        obs = obs.reshape(-1, self.obs_dim); act = act.reshape(-1, self.act_dim)
        rew = rew.reshape(-1, 1); nobs = nobs.reshape(-1, self.obs_dim); done = done.reshape(-1, 1) != 0
        copies, self.next_p, self.if_full, self.cur_capacity = ring_plan(self.next_p, self.if_full, self.capacity, rew.shape[0])
        for d, s, n in copies:
            self.obs[d:d + n] = obs[s:s + n]; self.act[d:d + n] = act[s:s + n]; self.rew[d:d + n] = rew[s:s + n]
            self.nobs[d:d + n] = nobs[s:s + n]; self.done[d:d + n] = done[s:s + n]

    def gather(self, idx):
        """sample_batch with the index vector supplied (simple_replay.py:98-104): done comes back fp32."""
        return self.obs[idx], self.act[idx], self.rew[idx], self.nobs[idx], self.done[idx].float()

    def sample(self, batch, generator=None):
        idx = torch.randint(self.cur_capacity, size=(batch,), generator=generator)  # :87
        return idx, self.gather(idx)


class ObsRingRef:
    """Obs-only ring of the P-learner (pql_p_learner.py:32-37, 66-85)."""

    def __init__(self, capacity: int, obs_dim: int):
        self.capacity, self.obs_dim = int(capacity), int(obs_dim)
        self.next_p, self.if_full, self.cur_capacity = 0, False, 0
        self.mem = torch.zeros((self.capacity, obs_dim))

    def insert(self, obs):
        obs = obs.reshape(-1, self.obs_dim)
        copies, self.next_p, self.if_full, self.cur_capacity = ring_plan(self.next_p, self.if_full, self.capacity, obs.shape[0])
        for d, s, n in copies:
            self.mem[d:d + n] = obs[s:s + n]

    def gather(self, idx):
        return self.mem[idx].clone()


# =========================================================================== n-step assembler
class NStepRef:
    """n-step transition assembler (pql/replay/nstep_replay.py:6-92), stated per emitted window
    instead of as FIFO tensor shifts.  Window slot j=0 is the oldest step.

    emit (nstep_replay.py:74-92): any = any(done_j != 0); first = index of the first maximal done_j;
    done_out = done_{n-1} or any;  next_obs_out = next_obs_{first} if any else next_obs_{n-1};
    R = sum_j (r_j * gamma^j) * [j <= first or not any], fp32, in the order given in _emit,
    gamma^j evaluated in double and rounded once to fp32 (:24).
    Output rows are time-major blocks of N (:65 torch.cat of per-step blocks).
    """

    def __init__(self, obs_dim: int, act_dim: int, num_envs: int, nstep: int = 3, gamma: float = 0.99):
        self.N, self.n, self.O, self.A = num_envs, nstep, obs_dim, act_dim
        self.gamma_pow = torch.tensor([gamma ** j for j in range(nstep)], dtype=torch.float64).to(torch.float32)
        self.count = 0
        self.w_obs = torch.zeros((num_envs, nstep, obs_dim)); self.w_act = torch.zeros((num_envs, nstep, act_dim))
        self.w_rew = torch.zeros((num_envs, nstep)); self.w_nobs = torch.zeros((num_envs, nstep, obs_dim))
        self.w_done = torch.zeros((num_envs, nstep))
        self.head = 0  # circular: slot of the OLDEST entry once full

    def _push(self, obs, act, rew, nobs, done):
        s = self.head  # overwrite the oldest slot, then the next one becomes oldest
        self.w_obs[:, s] = obs; self.w_act[:, s] = act; self.w_rew[:, s] = rew; self.w_nobs[:, s] = nobs; self.w_done[:, s] = done
        self.head = (self.head + 1) % self.n
        self.count += 1

    def _emit(self):
        order = [(self.head + j) % self.n for j in range(self.n)]  # oldest .. newest
        done = self.w_done[:, order]                    # (N, n)
        rew = self.w_rew[:, order]
        anyd = (done != 0).any(dim=1)
        mx = done.max(dim=1, keepdim=True).values
        first = (done == mx).float().argmax(dim=1)      # first index attaining the max
        j = torch.arange(self.n).unsqueeze(0)
        keep = torch.where(anyd.unsqueeze(1), j <= first.unsqueeze(1), torch.ones_like(j, dtype=torch.bool))
        # fp32 accumulation order of torch's contiguous row-sum for n <= 5 (what the reference's
        # `.sum(1)` does, :91): four interleaved partial sums A_i = sum_{j = i mod 4}, then
        # ((A0 + A1) + A2) + A3.  For n <= 4 that is plain left-to-right.
        lanes = [None] * 4
        for jj in range(self.n):
            term = (rew[:, jj] * self.gamma_pow[jj]) * keep[:, jj].float()
            lanes[jj % 4] = term if lanes[jj % 4] is None else lanes[jj % 4] + term
        R = lanes[0]
        for a in lanes[1:]:
            if a is not None:
                R = R + a
        sel = torch.where(anyd, first, torch.full_like(first, self.n - 1))
        slot = torch.tensor(order)[sel]
        ar = torch.arange(self.N)
        nobs = self.w_nobs[ar, slot]
        dout = torch.where(anyd, torch.ones(self.N), done[:, -1])
        return self.w_obs[:, order[0]].clone(), self.w_act[:, order[0]].clone(), R.unsqueeze(1), nobs.clone(), dout.unsqueeze(1)

    def add(self, obs, act, rew, nobs, done):
        """(N,T,.) slabs in -> 5 tensors out; nstep==1 is a pass-through of the inputs (:66-67)."""
        if self.n == 1:
            return obs, act, rew, nobs, done
        outs = [[], [], [], [], []]
        for t in range(obs.shape[1]):
            self._push(obs[:, t], act[:, t], rew[:, t].reshape(-1), nobs[:, t], done[:, t].reshape(-1))
            if self.count < self.n:
                continue
            for lst, v in zip(outs, self._emit()):
                lst.append(v)
        return tuple(torch.cat(l) for l in outs)  # raises on an empty list, as the reference does (:65)


# =========================================================================== small math
def normalize_ref(x, norm: Optional[Tuple[torch.Tensor, torch.Tensor, float]], clamp: bool = True):
    """Learner-side normalisation, clamp +-5 (pql/utils/common.py:139-145); clamp=False is the
    actor-side RunningMeanStd.normalize (pql/utils/torch_util.py:83-85)."""
    if norm is None:
        return x
    mean, var, eps = norm
    # sqrt through numpy: IEEE correctly rounded on every host.  torch's vectorised CPU sqrt is not
    # (about 0.6 % of inputs are 1 ulp off on some ISAs), which would make this oracle host-dependent;
    # the GPU's sqrtf and '/' are correctly rounded (tools/probes/probe_fp.hip).
    std = torch.from_numpy(np.sqrt((var.float() + eps).numpy()))
    y = (x - mean.float()) / std
    return y.clamp(-5.0, 5.0) if clamp else y


class RunningMeanStdRef:
    """Chan parallel-variance merge (pql/utils/torch_util.py:68-114): unbiased batch var, count starts at eps."""

    def __init__(self, shape, eps: float = 1e-4):
        self.mean = torch.zeros(shape); self.var = torch.ones(shape); self.eps = eps; self.count = eps

    def update(self, x):
        bm, bv, bc = x.mean(dim=0), x.var(dim=0), x.shape[0]
        delta = bm - self.mean
        tot = self.count + bc
        m2 = self.var * self.count + bv * bc + delta ** 2 * self.count * bc / tot
        self.mean = self.mean + delta * bc / tot
        self.var = m2 / tot
        self.count = tot

    def states(self):
        return self.mean, self.var, self.eps


def target_noise_ref(action, draw, std: float, bound: float):
    """a' = clamp(a + clamp(std*draw, +-bound), +-1)   (pql/utils/noise.py:19-27; draw ~ N(0,1))."""
    noise = (draw * std).clamp(-bound, bound)
    return (action + noise).clamp(-1.0, 1.0)


def mixed_noise_ref(action, draw, std_min: float, std_max: float):
    """Per-env sigma = linspace(std_min, std_max, N) (pql/utils/noise.py:30-41), no noise clamp."""
    std = torch.linspace(std_min, std_max, action.shape[0]).unsqueeze(-1)
    return (action + draw * std).clamp(-1.0, 1.0)


def c51_project_ref(p, reward, done, gamma_n: float, v_min: float, v_max: float, K: int):
    """Categorical projection (pql/utils/distl_util.py:4-20) written as an explicit per-atom loop with
    in-row accumulation: first all lower-neighbour deposits (atom order), then all upper ones,
    matching the two index_add_ passes of the reference."""
    dz = (v_max - v_min) / (K - 1)
    z = torch.linspace(v_min, v_max, K)
    tz = (reward + (1 - done) * gamma_n * z).clamp(min=v_min, max=v_max)   # (B,K)
    b = (tz - v_min) / dz
    lo = b.floor().long(); up = b.ceil().long()
    lo = torch.where((up > 0) & (lo == up), lo - 1, lo)
    up = torch.where((lo < K - 1) & (lo == up), up + 1, up)
    out = torch.zeros_like(p)
    w_lo = p * (up.float() - b)
    w_up = p * (b - lo.float())
    B = p.shape[0]
    rows = torch.arange(B)
    for k in range(K):
        out[rows, lo[:, k]] += w_lo[:, k]
    for k in range(K):
        out[rows, up[:, k]] += w_up[:, k]
    return out


# =========================================================================== MLP family
def layer_dims(in_dim: int, out_dim: int, hidden: Sequence[int] = HIDDEN_DEFAULT):
    d = [in_dim, *hidden, out_dim]
    return list(zip(d[:-1], d[1:]))


def mlp_forward_ref(params: List[torch.Tensor], x):
    """Linear->ELU x(L-1) -> Linear (pql/models/mlp.py:15-24); params = [W0,b0,W1,b1,...], W is (out,in)."""
    L = len(params) // 2
    for i in range(L):
        x = F.linear(x, params[2 * i], params[2 * i + 1])
        if i < L - 1:
            x = F.elu(x)
    return x


def params_from_state(state: Dict[str, np.ndarray], prefix: str = "net.") -> List[torch.Tensor]:
    out, i = [], 0
    while f"{prefix}{2 * i}.weight" in state:
        out.append(torch.as_tensor(np.array(state[f"{prefix}{2 * i}.weight"])).clone())
        out.append(torch.as_tensor(np.array(state[f"{prefix}{2 * i}.bias"])).clone())
        i += 1
    return out


def actor_forward_ref(params, obs):
    """TanhMLPPolicy (mlp.py:177-179)."""
    return torch.tanh(mlp_forward_ref(params, obs))


def twin_forward_ref(q1, q2, obs, act):
    """DoubleQ.get_q1_q2 (mlp.py:197-199)."""
    x = torch.cat((obs, act), dim=1)
    return mlp_forward_ref(q1, x), mlp_forward_ref(q2, x)


def twin_dist_ref(q1, q2, obs, act):
    """DistributionalDoubleQ.get_q1_q2 (mlp.py:261-263): softmax over atoms."""
    l1, l2 = twin_forward_ref(q1, q2, obs, act)
    return torch.softmax(l1, dim=1), torch.softmax(l2, dim=1)


def qmin_ref(q1, q2, obs, act, z_atoms=None):
    """get_q_min: (B,1) for DoubleQ (mlp.py:194-195); (B,) expectation-min for the distributional net (:256-260)."""
    if z_atoms is None:
        a, b = twin_forward_ref(q1, q2, obs, act)
        return torch.min(a, b)
    a, b = twin_dist_ref(q1, q2, obs, act)
    return torch.min((a * z_atoms).sum(dim=1), (b * z_atoms).sum(dim=1))


# =========================================================================== optimiser
@dataclass
class AdamWRef:
    """clip_grad_norm_ + torch.optim.AdamW defaults + optional Polyak, over a parameter list
    (pql_v_learner.py:124-133, pql_p_learner.py:87-96, torch_util.py:9-12)."""
    params: List[torch.Tensor]
    lr: float = 5e-4
    b1: float = 0.9
    b2: float = 0.999
    eps: float = 1e-8
    wd: float = 1e-2
    step: int = 0
    m: List[torch.Tensor] = field(default_factory=list)
    v: List[torch.Tensor] = field(default_factory=list)

    def __post_init__(self):
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]

    @torch.no_grad()
    def apply(self, grads: List[torch.Tensor], max_norm: Optional[float]):
        if max_norm is not None:
            total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads]))
            coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
            grads = [g * coef for g in grads]
        self.step += 1
        bc1 = 1 - self.b1 ** self.step
        bc2 = 1 - self.b2 ** self.step
        for p, g, m, v in zip(self.params, grads, self.m, self.v):
            p.mul_(1 - self.lr * self.wd)
            m.lerp_(g, 1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(m, denom, value=-(self.lr / bc1))


@torch.no_grad()
def polyak_ref(target: List[torch.Tensor], current: List[torch.Tensor], tau: float):
    for t, c in zip(target, current):
        t.copy_(c * tau + t * (1.0 - tau))


# =========================================================================== learners
@dataclass
class HyperRef:
    batch_size: int = 8192
    gamma: float = 0.99
    nstep: int = 3
    tau: float = 0.05
    max_grad_norm: Optional[float] = 0.5
    critic_lr: float = 5e-4
    actor_lr: float = 5e-4
    obs_norm: bool = True
    distl: bool = False
    v_min: float = -10.0
    v_max: float = 10.0
    num_atoms: int = 51
    tgt_pol_std: float = 0.8
    tgt_pol_noise_bound: float = 0.2


class VLearnerRef:
    """One critic gradient step = PQLVLearner.learn (pql/algo/pql_v_learner.py:73-115)."""

    def __init__(self, obs_dim, act_dim, hp: HyperRef, capacity: int, q1: List[torch.Tensor], q2: List[torch.Tensor]):
        self.hp, self.O, self.A = hp, obs_dim, act_dim
        self.q1 = [p.clone().requires_grad_(True) for p in q1]
        self.q2 = [p.clone().requires_grad_(True) for p in q2]
        self.t1 = [p.detach().clone() for p in self.q1]
        self.t2 = [p.detach().clone() for p in self.q2]
        self.opt = AdamWRef([*self.q1, *self.q2], lr=hp.critic_lr)
        self.ring = RingRef(capacity, obs_dim, act_dim)
        self.actor: Optional[List[torch.Tensor]] = None
        self.norm = None
        self.update_count = 0
        self.z = torch.linspace(hp.v_min, hp.v_max, hp.num_atoms) if hp.distl else None

    def update(self, actor_params, traj, norm):
        self.actor = [p.detach().clone() for p in actor_params]
        self.ring.insert(*traj)
        self.norm = norm

    def learn(self, idx=None, draw=None, generator=None):
        if self.actor is None:
            return None
        loss, grads = self.loss_and_grads(idx, draw, generator)
        params = [*self.q1, *self.q2]
        self.opt.apply(list(grads), self.hp.max_grad_norm)
        polyak_ref([*self.t1, *self.t2], [p.detach() for p in params], self.hp.tau)
        self.update_count += 1
        return float(loss.detach())

    def loss_and_grads(self, idx=None, draw=None, generator=None):
        """The first half of learn(): critic loss and its gradient (what a data-parallel rank all-reduces)."""
        hp = self.hp
        if idx is None:
            idx = torch.randint(self.ring.cur_capacity, size=(hp.batch_size,), generator=generator)
        obs, act, rew, nobs, done = self.ring.gather(idx)
        if hp.obs_norm:
            obs = normalize_ref(obs, self.norm); nobs = normalize_ref(nobs, self.norm)
        gn = hp.gamma ** hp.nstep
        with torch.no_grad():
            na = actor_forward_ref(self.actor, nobs)
            if draw is None:
                draw = torch.empty_like(na).normal_(generator=generator)
            na = target_noise_ref(na, draw, hp.tgt_pol_std, hp.tgt_pol_noise_bound)
            if hp.distl:
                p1, p2 = twin_dist_ref(self.t1, self.t2, nobs, na)
                tgt = torch.min(c51_project_ref(p1, rew, done, gn, hp.v_min, hp.v_max, hp.num_atoms),
                                c51_project_ref(p2, rew, done, gn, hp.v_min, hp.v_max, hp.num_atoms))
            else:
                tgt = rew + (1 - done) * gn * qmin_ref(self.t1, self.t2, nobs, na)
        if hp.distl:
            c1, c2 = twin_dist_ref(self.q1, self.q2, obs, act)
            loss = F.binary_cross_entropy(c1, tgt) + F.binary_cross_entropy(c2, tgt)
        else:
            c1, c2 = twin_forward_ref(self.q1, self.q2, obs, act)
            loss = F.mse_loss(c1, tgt) + F.mse_loss(c2, tgt)
        return loss, torch.autograd.grad(loss, [*self.q1, *self.q2])


class PLearnerRef:
    """One DPG actor step = PQLPLearner.learn (pql/algo/pql_p_learner.py:47-64)."""

    def __init__(self, obs_dim, act_dim, hp: HyperRef, capacity: int, actor: List[torch.Tensor]):
        self.hp, self.O, self.A = hp, obs_dim, act_dim
        self.actor = [p.clone().requires_grad_(True) for p in actor]
        self.opt = AdamWRef(self.actor, lr=hp.actor_lr)
        self.ring = ObsRingRef(capacity, obs_dim)
        self.q1 = self.q2 = None
        self.norm = None
        self.update_count = 0
        self.z = torch.linspace(hp.v_min, hp.v_max, hp.num_atoms) if hp.distl else None

    def update(self, q1, q2, obs, norm):
        self.q1 = [p.detach().clone() for p in q1]; self.q2 = [p.detach().clone() for p in q2]
        self.ring.insert(obs)
        self.norm = norm

    def learn(self, idx=None, generator=None):
        hp = self.hp
        if self.q1 is None:
            return None
        if idx is None:
            idx = torch.randint(self.ring.cur_capacity, size=(hp.batch_size,), generator=generator)
        obs = self.ring.gather(idx)
        if hp.obs_norm:
            obs = normalize_ref(obs, self.norm)
        a = actor_forward_ref(self.actor, obs)
        loss = -qmin_ref(self.q1, self.q2, obs, a, self.z).mean()
        grads = torch.autograd.grad(loss, self.actor)
        self.opt.apply(list(grads), hp.max_grad_norm)
        self.update_count += 1
        return float(loss.detach())


# =========================================================================== DDPG (BASELINE cfg #1)
class DDPGRef:
    """AgentDDPG.update_net inner iteration (pql/algo/ddpg.py:119-166): shared batch, critic step (MSE TD), actor step (DPG),
    Polyak on the critic and -- `actor_target` given, i.e. `no_tgt_actor=False` (ddpg.py:21-22,134-135) -- on the target actor,
    which then supplies the target-policy actions (ddpg.py:70-79); without it the target actor IS the actor.
    Normalisation here is the un-clamped RunningMeanStd.normalize (ddpg.py:124-126).
    Pinned by tests/golden/ddpg.npz (the reference's own AgentDDPG, both settings)."""

    def __init__(self, obs_dim, act_dim, hp: HyperRef, capacity, actor, q1, q2, actor_target=None):
        self.hp = hp
        self.actor = [p.clone().requires_grad_(True) for p in actor]
        self.actor_t = [p.detach().clone() for p in actor_target] if actor_target is not None else None
        self.q1 = [p.clone().requires_grad_(True) for p in q1]
        self.q2 = [p.clone().requires_grad_(True) for p in q2]
        self.t1 = [p.detach().clone() for p in self.q1]; self.t2 = [p.detach().clone() for p in self.q2]
        self.aopt = AdamWRef(self.actor, lr=hp.actor_lr)
        self.copt = AdamWRef([*self.q1, *self.q2], lr=hp.critic_lr)
        self.ring = RingRef(capacity, obs_dim, act_dim)
        self.norm = None

    def update_once(self, idx, draw):
        hp = self.hp
        obs, act, rew, nobs, done = self.ring.gather(idx)
        if hp.obs_norm:
            obs = normalize_ref(obs, self.norm, clamp=False); nobs = normalize_ref(nobs, self.norm, clamp=False)
        with torch.no_grad():
            tgt_actor = self.actor_t if self.actor_t is not None else self.actor
            na = target_noise_ref(actor_forward_ref(tgt_actor, nobs), draw, hp.tgt_pol_std, hp.tgt_pol_noise_bound)
            tgt = rew + (1 - done) * (hp.gamma ** hp.nstep) * qmin_ref(self.t1, self.t2, nobs, na)
        c1, c2 = twin_forward_ref(self.q1, self.q2, obs, act)
        closs = F.mse_loss(c1, tgt) + F.mse_loss(c2, tgt)
        cp = [*self.q1, *self.q2]
        self.copt.apply(list(torch.autograd.grad(closs, cp)), hp.max_grad_norm)
        frozen1 = [p.detach() for p in self.q1]; frozen2 = [p.detach() for p in self.q2]
        aloss = -qmin_ref(frozen1, frozen2, obs, actor_forward_ref(self.actor, obs)).mean()
        self.aopt.apply(list(torch.autograd.grad(aloss, self.actor)), hp.max_grad_norm)
        polyak_ref([*self.t1, *self.t2], [p.detach() for p in cp], hp.tau)
        if self.actor_t is not None:
            polyak_ref(self.actor_t, [p.detach() for p in self.actor], hp.tau)
        return float(closs.detach()), float(aloss.detach())


# ------------------------------------------------------------------------------------------------ SAC
LOG_SQRT_2PI = math.log(math.sqrt(2 * math.pi))


def squashed_gaussian_ref(params, obs, eps):
    """TanhDiagGaussianMLPPolicy.get_actions_logprob (pql/models/mlp.py:144-174) with the draw `eps` of Normal.rsample
    supplied: SquashedNormal = TanhTransform(cache_size=1) over Normal(mu, exp(clamp(log_std, -5, 5)))
    (pql/utils/torch_util.py:15-65; torch.distributions.Normal.rsample / log_prob).  Returns (a (B,A), logp (B,1))."""
    mu, log_std = mlp_forward_ref(params, obs).chunk(2, dim=-1)
    std = log_std.clamp(-5, 5).exp()
    u = mu + eps * std
    a = u.tanh()
    jac = 2.0 * (math.log(2.0) - u - F.softplus(-2.0 * u))          # TanhTransform.log_abs_det_jacobian on the cached u
    base = -((u - mu) ** 2) / (2 * std ** 2) - std.log() - LOG_SQRT_2PI
    return a, ((0.0 - jac) + base).sum(-1, keepdim=True)


class SACRef:
    """AgentSAC.update_net inner iteration (pql/algo/sac.py:98-108,138-156), no_tgt_actor=True, learned temperature:
    shared batch (normalised without clamp), critic step on the entropy-regularised n-step target, actor step through
    the UPDATED critic, temperature step, Polyak on the critic."""

    def __init__(self, obs_dim, act_dim, hp: HyperRef, capacity, actor, q1, q2, alpha_lr=5e-3, alpha=None):
        self.hp = hp
        self.actor = [p.clone().requires_grad_(True) for p in actor]
        self.q1 = [p.clone().requires_grad_(True) for p in q1]
        self.q2 = [p.clone().requires_grad_(True) for p in q2]
        self.t1 = [p.detach().clone() for p in self.q1]; self.t2 = [p.detach().clone() for p in self.q2]
        self.aopt = AdamWRef(self.actor, lr=hp.actor_lr)
        self.copt = AdamWRef([*self.q1, *self.q2], lr=hp.critic_lr)
        self.fixed_alpha = alpha
        self.log_alpha = torch.zeros(1, requires_grad=True)
        self.alpha_opt = AdamWRef([self.log_alpha], lr=alpha_lr)
        self.target_entropy = -act_dim
        self.ring = RingRef(capacity, obs_dim, act_dim)
        self.norm = None

    def alpha(self):
        return self.log_alpha.detach().exp() if self.fixed_alpha is None else self.fixed_alpha

    def update_once(self, idx, eps_next, eps_cur):
        hp = self.hp
        obs, act, rew, nobs, done = self.ring.gather(idx)
        if hp.obs_norm:
            obs = normalize_ref(obs, self.norm, clamp=False); nobs = normalize_ref(nobs, self.norm, clamp=False)
        with torch.no_grad():
            na, nlogp = squashed_gaussian_ref(self.actor, nobs, eps_next)
            tq = qmin_ref(self.t1, self.t2, nobs, na) - self.alpha() * nlogp
            tgt = rew + (1 - done) * (hp.gamma ** hp.nstep) * tq
        c1, c2 = twin_forward_ref(self.q1, self.q2, obs, act)
        closs = F.mse_loss(c1, tgt) + F.mse_loss(c2, tgt)
        cp = [*self.q1, *self.q2]
        self.copt.apply(list(torch.autograd.grad(closs, cp)), hp.max_grad_norm)
        frozen1 = [p.detach() for p in self.q1]; frozen2 = [p.detach() for p in self.q2]
        a, logp = squashed_gaussian_ref(self.actor, obs, eps_cur)
        aloss = (self.alpha() * logp - qmin_ref(frozen1, frozen2, obs, a)).mean()
        self.aopt.apply(list(torch.autograd.grad(aloss, self.actor)), hp.max_grad_norm)
        alpha_loss = None
        if self.fixed_alpha is None:
            alpha_loss = (self.log_alpha.exp() * (-logp - self.target_entropy).detach()).mean()
            self.alpha_opt.apply(list(torch.autograd.grad(alpha_loss, [self.log_alpha])), hp.max_grad_norm)
        polyak_ref([*self.t1, *self.t2], [p.detach() for p in cp], hp.tau)
        return float(closs.detach()), float(aloss.detach()), None if alpha_loss is None else float(alpha_loss.detach())


# ------------------------------------------------------------------------------------------------ CrossQ
def bn_mlp_forward_ref(lin, bn, stats, x, training=True, momentum=0.1, eps=1e-5):
    """create_simple_mlp(use_batchnorm=True) (pql/models/mlp.py:15-24): Linear -> BatchNorm1d -> ELU per hidden layer.
    lin = [W0, b0, W1, b1, ...]; bn = [gamma0, beta0, ...]; stats = [running_mean0, running_var0, ...] (updated in place
    in training mode, exactly as nn.BatchNorm1d does: momentum 0.1, unbiased variance)."""
    n_layers = len(lin) // 2
    for l in range(n_layers):
        x = F.linear(x, lin[2 * l], lin[2 * l + 1])
        if l < n_layers - 1:
            x = F.elu(F.batch_norm(x, stats[2 * l], stats[2 * l + 1], bn[2 * l], bn[2 * l + 1], training, momentum, eps))
    return x


class CrossQRef:
    """AgentCrossQ.update_net inner iteration (pql/algo/crossQ.py:120-166), no_tgt_actor=True: joint BatchNorm-critic
    forward over [obs; next_obs], target from the detached next half, twin MSE; DPG actor step through the critic in
    training mode; no target networks."""

    def __init__(self, obs_dim, act_dim, hp: HyperRef, capacity, actor, q_lin, q_bn):
        """q_lin[n] / q_bn[n]: Linear and BatchNorm parameter lists of net n."""
        self.hp = hp
        self.actor = [p.clone().requires_grad_(True) for p in actor]
        self.q_lin = [[p.clone().requires_grad_(True) for p in net] for net in q_lin]
        self.q_bn = [[p.clone().requires_grad_(True) for p in net] for net in q_bn]
        self.q_stats = []
        for net in q_bn:
            st = []
            for g in net[0::2]:
                st += [torch.zeros_like(g), torch.ones_like(g)]
            self.q_stats.append(st)
        self.aopt = AdamWRef(self.actor, lr=hp.actor_lr)
        self.cparams = [*self.q_lin[0], *self.q_bn[0], *self.q_lin[1], *self.q_bn[1]]
        self.copt = AdamWRef(self.cparams, lr=hp.critic_lr)
        self.ring = RingRef(capacity, obs_dim, act_dim)
        self.norm = None

    def q12(self, obs, act, lin=None, bn=None):
        x = torch.cat((obs, act), dim=1)
        lin, bn = lin or self.q_lin, bn or self.q_bn
        return tuple(bn_mlp_forward_ref(lin[n], bn[n], self.q_stats[n], x, training=True) for n in range(2))

    def update_once(self, idx, draw):
        hp = self.hp
        obs, act, rew, nobs, done = self.ring.gather(idx)
        if hp.obs_norm:
            obs = normalize_ref(obs, self.norm, clamp=False); nobs = normalize_ref(nobs, self.norm, clamp=False)
        with torch.no_grad():
            na = target_noise_ref(actor_forward_ref(self.actor, nobs), draw, hp.tgt_pol_std, hp.tgt_pol_noise_bound)
        a1, a2 = self.q12(torch.cat((obs, nobs), dim=0), torch.cat((act, na), dim=0))
        B = obs.shape[0]
        tgt = rew + (1 - done) * (hp.gamma ** hp.nstep) * torch.min(a1[B:].detach(), a2[B:].detach())
        closs = F.mse_loss(a1[:B], tgt) + F.mse_loss(a2[:B], tgt)
        self.copt.apply(list(torch.autograd.grad(closs, self.cparams)), hp.max_grad_norm)
        flin = [[p.detach() for p in net] for net in self.q_lin]; fbn = [[p.detach() for p in net] for net in self.q_bn]
        q1, q2 = self.q12(obs, actor_forward_ref(self.actor, obs), flin, fbn)
        aloss = -torch.min(q1, q2).mean()
        self.aopt.apply(list(torch.autograd.grad(aloss, self.actor)), hp.max_grad_norm)
        return float(closs.detach()), float(aloss.detach())


# --------------------------------------------------------------------------------------------------------------
# a25: the speed-ratio control law of the main loop (scripts/train_pql.py:72-88 state, :127-166 law + bookkeeping).
# Plain-Python restatement used to check pql_amd.utils.ratio_control.RatioController; "parity unpinned" beyond that:
# the reference has no fixture for it and its inputs are wall-clock times, so the check is law-vs-law on a
# scripted sequence of (time, critic count, actor count) observations.
def ratio_control_ref(observations, critic_sample_ratio=8, critic_actor_ratio=2, first=(0.0, 0, 0)):
    """observations: [(time at the end of rollout iteration i, critic_update_times, actor_update_times)], i = 1..n;
    `first` = (time, critic, actor) of the entry made before the loop (:79-85).  Returns the (sim_wait_time,
    critic_wait_time, actor_wait_time) in force after each iteration."""
    from collections import deque
    window = deque(maxlen=100)                                                   # :77-78
    waits = dict(sim=0, critic=0, actor=0)                                       # :73-76
    window.append(dict(time=first[0], sim=0, critic=first[1], actor=first[2], **{k + "_w": v for k, v in waits.items()}))
    out = []
    for sim_count, (now, n_cri, n_act) in enumerate(observations, start=1):      # :97-99
        if len(window) >= 10 and n_act != 0 and n_cri != 0:                      # :127
            head = window[0]
            dt = now - head["time"]                                              # :128
            try:
                sim_u, cri_u, act_u = dt / (sim_count - head["sim"]), dt / (n_cri - head["critic"]), dt / (n_act - head["actor"])
            except ZeroDivisionError:                                            # the reference would raise here; the build skips
                sim_u = None
            if sim_u is not None:
                w = sim_u / critic_sample_ratio - cri_u                          # :133
                if w > 0:
                    if waits["sim"] == 0:
                        waits["critic"] = head["critic_w"] + w                  # :136
                    else:
                        waits["sim"] = max(0, head["sim_w"] - w)                # :138
                else:
                    if waits["critic"] == 0:
                        waits["sim"] = head["sim_w"] - w                        # :141
                    else:
                        waits["critic"] = max(0, head["critic_w"] + w)          # :143
                w = cri_u * critic_actor_ratio - act_u                           # :145
                waits["actor"] = head["actor_w"] + w if w > 0 else max(0, head["actor_w"] + w)   # :146-149
        window.append(dict(time=now, sim=sim_count, critic=n_cri, actor=n_act, **{k + "_w": v for k, v in waits.items()}))   # :151-157
        out.append((waits["sim"], waits["critic"], waits["actor"]))
    return out
