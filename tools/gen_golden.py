#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the *reference* PQL code (read-only at
/root/reference) on CPU.  Runs only in the build container; the outputs are
plain arrays (inputs are re-derivable from tests/detdata.py seeds, so fixtures
hold expected outputs plus the captured RNG draws).  No reference source or
bytecode is copied anywhere.

Stand-ins for packages the container lacks (gym, wandb, loguru, omegaconf,
escnn, ray) are empty shells registered in sys.modules before the import, as
recorded in SURVEY.md Appendix A; none of them contributes arithmetic.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py
"""
import contextlib
import os
import sys
import types
from copy import deepcopy
from types import SimpleNamespace as NS

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import detdata as dd  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def _install_shells():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Empty:
        pass

    spaces = mod("gym.spaces", Discrete=_Empty, Box=_Empty, Dict=_Empty)
    mod("gym", spaces=spaces)
    mod("wandb")
    quiet = NS(info=lambda *a, **k: None, warning=lambda *a, **k: None, error=lambda *a, **k: None)
    mod("loguru", logger=quiet)
    mod("omegaconf", OmegaConf=_Empty, open_dict=contextlib.nullcontext, DictConfig=dict)
    mod("omegaconf.dictconfig", DictConfig=dict)
    enn = mod("escnn.nn", FieldType=_Empty)
    mod("escnn", nn=enn)

    def remote(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda obj: obj

    mod("ray", remote=remote, get=lambda x: x)
    sym = mod("bidex.utils.symmetry", load_symmetric_system=None)   # imported by pql/algo/ac_base.py:12, never called here
    mod("bidex.utils", symmetry=sym)
    mod("bidex")


def _import_reference():
    _install_shells()
    sys.path.insert(0, REF)
    import pql.models  # noqa: F401  (package __init__ only scans file names)
    emlp = types.ModuleType("pql.models.emlp")
    emlp.EMLP = emlp.EMLPNew = type("EMLP", (), {})
    sys.modules["pql.models.emlp"] = emlp
    from pql.replay.simple_replay import ReplayBuffer
    from pql.replay.nstep_replay import NStepReplay
    from pql.models.mlp import TanhMLPPolicy, DoubleQ, DistributionalDoubleQ, MLPNet
    from pql.utils.distl_util import projection
    from pql.utils.common import normalize, Tracker
    from pql.utils.torch_util import soft_update, RunningMeanStd
    from pql.utils.noise import add_normal_noise, add_mixed_normal_noise
    from pql.algo.pql_v_learner import PQLVLearner
    from pql.algo.pql_p_learner import PQLPLearner
    from pql.models.mlp import TanhDiagGaussianMLPPolicy
    from pql.algo.sac import AgentSAC
    from pql.models.mlp import DoubleQBatchNorm
    from pql.algo.crossQ import AgentCrossQ
    from pql.algo.ddpg import AgentDDPG
    return NS(**locals())


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def load_state(module, state):
    module.load_state_dict({k: T(v) for k, v in state.items()})


# --------------------------------------------------------------------------- ring buffer
def gen_ring(R, out):
    """a3/a4/a5: pointer traces incl. exact fill and wrap; gathers for fixed indices."""
    cases = [
        ("wrap4", 10, 3, 2, [4, 4, 4, 4]),          # SURVEY §4 KAT: next_p 4,8,2,6
        ("exact", 8, 5, 1, [4, 4, 4]),              # p == capacity (no wrap branch), then wrap from p=cap
        ("ragged", 13, 8, 2, [5, 1, 7, 13, 2, 12]),  # insert == capacity, odd sizes
    ]
    for name, cap, O, A, inserts in cases:
        rb = R.ReplayBuffer(capacity=cap, obs_dim=(O,), action_dim=A, device="cpu")
        # make uninitialised storage deterministic so full-buffer snapshots are comparable
        rb.buf_obs.zero_(); rb.buf_action.zero_(); rb.buf_next_obs.zero_(); rb.buf_reward.zero_(); rb.buf_done.zero_()
        trace = []
        for step, m in enumerate(inserts):
            seed = 100 + step
            traj = (T(dd.uniform((m, O), seed)), T(dd.uniform((m, A), seed + 20)), T(dd.uniform((m, 1), seed + 40)),
                    T(dd.uniform((m, O), seed + 60)), T(dd.bernoulli((m, 1), seed + 80, 0.3)))
            rb.add_to_buffer(traj)
            trace.append([rb.next_p, rb.cur_capacity, int(rb.if_full)])
        out[f"ring_{name}_meta"] = np.array([cap, O, A], dtype=np.int64)
        out[f"ring_{name}_inserts"] = np.array(inserts, dtype=np.int64)
        out[f"ring_{name}_trace"] = np.array(trace, dtype=np.int64)
        out[f"ring_{name}_obs"] = rb.buf_obs.numpy().copy()
        out[f"ring_{name}_act"] = rb.buf_action.numpy().copy()
        out[f"ring_{name}_rew"] = rb.buf_reward.numpy().copy()
        out[f"ring_{name}_nobs"] = rb.buf_next_obs.numpy().copy()
        out[f"ring_{name}_done"] = rb.buf_done.numpy().copy()
        idx = dd.integers((16,), 900, rb.cur_capacity)
        real_randint = torch.randint
        torch.randint = lambda *a, **k: T(idx)
        try:
            s = rb.sample_batch(16, device="cpu")
        finally:
            torch.randint = real_randint
        out[f"ring_{name}_idx"] = idx
        for nm, t in zip(("s_obs", "s_act", "s_rew", "s_nobs", "s_done"), s):
            out[f"ring_{name}_{nm}"] = t.numpy().copy()

    # a5: P-learner obs-only ring (same pointer law, inline in update())
    p = R.PQLPLearner.__new__(R.PQLPLearner)
    p.obs_dim, p.memory_size = (4,), 10
    p.memory = torch.zeros((10, 4)); p.next_p = 0; p.if_full = False; p.cur_capacity = 0
    p.actor = None; p.loss_tracker = R.Tracker(5); p.update_count = 0
    trace = []
    for step, m in enumerate([4, 4, 4, 4, 10]):
        p.update(None, T(dd.uniform((m, 4), 300 + step)), None, 0.0)
        trace.append([p.next_p, p.cur_capacity, int(p.if_full)])
    out["pring_trace"] = np.array(trace, dtype=np.int64)
    out["pring_mem"] = p.memory.numpy().copy()


# --------------------------------------------------------------------------- n-step
def gen_nstep(R, out):
    """a6/a7: first call T>=nstep then T=1 calls; dones at every window slot, multiple dones."""
    cases = [("kat3", 3, 3, 2, 1, [5]),                 # SURVEY §4 KAT geometry
             ("n3", 6, 3, 5, 2, [4, 1, 1, 3, 1]),
             ("n5", 7, 5, 3, 2, [32, 1, 1, 1, 1, 1, 2]),
             ("n1", 4, 1, 3, 2, [2, 1])]
    for name, N, n, O, A, calls in cases:
        ns = R.NStepReplay((O,), A, N, n, device="cpu")
        for ci, Tn in enumerate(calls):
            seed = 500 + 10 * ci
            obs = dd.uniform((N, Tn, O), seed); act = dd.uniform((N, Tn, A), seed + 1)
            rew = dd.uniform((N, Tn, 1), seed + 2); nobs = dd.uniform((N, Tn, O), seed + 3)
            done = dd.bernoulli((N, Tn, 1), seed + 4, 0.25)
            if name == "kat3":
                rew = np.ones_like(rew); done = np.zeros_like(done); done[1, 1] = 1; done[2, 4] = 1
            res = ns.add_to_buffer(T(obs), T(act), T(rew), T(nobs), T(done))
            for nm, t in zip(("obs", "act", "rew", "nobs", "done"), res):
                out[f"nstep_{name}_c{ci}_{nm}"] = t.to(torch.float32).numpy().copy()
            if name == "kat3":
                out["nstep_kat3_in_rew"] = rew; out["nstep_kat3_in_done"] = done
        out[f"nstep_{name}_meta"] = np.array([N, n, O, A] + calls, dtype=np.int64)
        out[f"nstep_{name}_gamma"] = ns.gamma_array.numpy().copy() if n > 1 else np.ones((1, 1), np.float32)


# --------------------------------------------------------------------------- models
MODEL_SHAPES = [(8, 2), (88, 16), (211, 20), (108, 21)]


def gen_models(R, out):
    """a8-a11: forward outputs and parameter/input gradients for fixed weights."""
    for (O, A) in MODEL_SHAPES:
        B = 33 if O == 8 else 17
        tag = f"o{O}a{A}"
        obs = T(dd.uniform((B, O), 1000 + O, -2, 2)); act = T(dd.uniform((B, A), 2000 + O))
        # actor
        actor = R.TanhMLPPolicy((O,), A)
        load_state(actor, dd.mlp_state(O, A, 11))
        o = obs.clone().requires_grad_(True)
        y = actor(o)
        w = T(dd.uniform((B, A), 3000 + O))
        (y * w).sum().backward()
        out[f"actor_{tag}_y"] = y.detach().numpy()
        out[f"actor_{tag}_dobs"] = o.grad.numpy()
        for k, p in actor.named_parameters():
            out[f"actor_{tag}_g_{k}"] = dd.summarize(p.grad.numpy())
        # DoubleQ
        q = R.DoubleQ((O,), A)
        load_state(q, dd.doubleq_state(O, A, 1, 21))
        o = obs.clone().requires_grad_(True); a = act.clone().requires_grad_(True)
        q1, q2 = q.get_q1_q2(o, a)
        out[f"dq_{tag}_q1"] = q1.detach().numpy(); out[f"dq_{tag}_q2"] = q2.detach().numpy()
        out[f"dq_{tag}_qmin"] = q.get_q_min(o, a).detach().numpy()
        tgt = T(dd.uniform((B, 1), 4000 + O))
        loss = torch.nn.functional.mse_loss(q1, tgt) + torch.nn.functional.mse_loss(q2, tgt)
        loss.backward()
        out[f"dq_{tag}_loss"] = np.array(loss.item(), np.float64)
        out[f"dq_{tag}_dobs"] = o.grad.numpy(); out[f"dq_{tag}_dact"] = a.grad.numpy()
        for k, p in q.named_parameters():
            out[f"dq_{tag}_g_{k}"] = dd.summarize(p.grad.numpy())
        # DPG path: -mean(min Q) wrt action and obs
        q.zero_grad()
        o = obs.clone().requires_grad_(True); a = act.clone().requires_grad_(True)
        (-q.get_q_min(o, a).mean()).backward()
        out[f"dq_{tag}_dpg_dact"] = a.grad.numpy(); out[f"dq_{tag}_dpg_dobs"] = o.grad.numpy()
        # DistributionalDoubleQ
        K = 51
        dq = R.DistributionalDoubleQ((O,), A, v_min=-10, v_max=10, num_atoms=K, device="cpu")
        load_state(dq, dd.doubleq_state(O, A, K, 31))
        o = obs.clone().requires_grad_(True); a = act.clone().requires_grad_(True)
        p1, p2 = dq.get_q1_q2(o, a)
        out[f"ddq_{tag}_p1"] = p1.detach().numpy(); out[f"ddq_{tag}_p2"] = p2.detach().numpy()
        out[f"ddq_{tag}_qmin"] = dq.get_q_min(o, a).detach().numpy()
        out[f"ddq_{tag}_z"] = dq.z_atoms.numpy()
        tg = T(dd.uniform((B, K), 5000 + O, 0.0, 1.0)); tg = tg / tg.sum(1, keepdim=True)
        out[f"ddq_{tag}_tgt"] = tg.numpy()
        loss = torch.nn.functional.binary_cross_entropy(p1, tg) + torch.nn.functional.binary_cross_entropy(p2, tg)
        loss.backward()
        out[f"ddq_{tag}_loss"] = np.array(loss.item(), np.float64)
        out[f"ddq_{tag}_dobs"] = o.grad.numpy(); out[f"ddq_{tag}_dact"] = a.grad.numpy()
        for k, p in dq.named_parameters():
            out[f"ddq_{tag}_g_{k}"] = dd.summarize(p.grad.numpy())
        dq.zero_grad()
        o = obs.clone().requires_grad_(True); a = act.clone().requires_grad_(True)
        (-dq.get_q_min(o, a).mean()).backward()
        out[f"ddq_{tag}_dpg_dact"] = a.grad.numpy()
    # BASELINE hidden shape through MLPNet(hidden_layers=...)
    hid = (512, 512, 256)
    net = R.MLPNet(104, 1, hidden_layers=list(hid))
    load_state(net, dd.mlp_state(104, 1, 41, hid))
    x = T(dd.uniform((9, 104), 6000)).requires_grad_(True)
    y = net(x); y.sum().backward()
    out["mlp_h512_512_256_y"] = y.detach().numpy(); out["mlp_h512_512_256_dx"] = x.grad.numpy()


# --------------------------------------------------------------------------- math utils
def gen_math(R, out):
    # a16 projection: integral b, r beyond +-v, done=1, generic
    K = 51
    B = 12
    p = dd.uniform((B, K), 700, 0.0, 1.0); p = (p / p.sum(1, keepdims=True)).astype(np.float32)
    rew = dd.uniform((B, 1), 701, -4, 4)
    done = dd.bernoulli((B, 1), 702, 0.4)
    rew[0, 0] = -3.0; done[0, 0] = 1.0      # SURVEY KAT: mass .5/.5 on atoms 17/18
    rew[1, 0] = 20.0; done[1, 0] = 1.0      # clamp high -> atom 50
    rew[2, 0] = -20.0; done[2, 0] = 0.0     # clamp low
    rew[3, 0] = 0.4; done[3, 0] = 1.0       # b integral (0.4 -> b = 26)
    rew[4, 0] = -10.0; done[4, 0] = 1.0     # b == 0 edge (l==u==0)
    rew[5, 0] = 10.0; done[5, 0] = 1.0      # b == K-1 edge
    z = torch.linspace(-10, 10, K)
    g = 0.99 ** 3
    pr = R.projection(T(p), T(rew), T(done), g, -10, 10, K, z, device="cpu")
    out["proj_p"] = p; out["proj_rew"] = rew; out["proj_done"] = done
    out["proj_out"] = pr.numpy(); out["proj_gamma"] = np.array(g, np.float64)
    # second geometry (v range not symmetric, K=11)
    z2 = torch.linspace(-2, 6, 11)
    p2 = dd.uniform((7, 11), 710, 0.0, 1.0); p2 = (p2 / p2.sum(1, keepdims=True)).astype(np.float32)
    r2 = dd.uniform((7, 1), 711, -3, 7); d2 = dd.bernoulli((7, 1), 712, 0.3)
    out["proj2_p"] = p2; out["proj2_rew"] = r2; out["proj2_done"] = d2
    out["proj2_out"] = R.projection(T(p2), T(r2), T(d2), 0.95, -2, 6, 11, z2, device="cpu").numpy()
    # a12 normalize (with clamp) and a13 RunningMeanStd
    x = dd.uniform((40, 6), 720, -30, 30)
    mean = dd.uniform((6,), 721); var = dd.uniform((6,), 722, 0.01, 4.0)
    out["norm_x"] = x; out["norm_mean"] = mean; out["norm_var"] = var
    out["norm_y"] = R.normalize(T(x), (T(mean), T(var), 1e-4)).numpy()
    rms = R.RunningMeanStd(shape=(6,), device="cpu")
    st = []
    for i in range(3):
        rms.update(T(dd.uniform((32, 6), 730 + i, -3, 5)))
        st.append(np.concatenate([rms.mean.numpy(), rms.var.numpy(), [rms.count]]))
    out["rms_trace"] = np.array(st, dtype=np.float64)
    out["rms_norm"] = rms.normalize(T(x)).numpy()
    # a14 noise with injected normal draw
    a = dd.uniform((10, 4), 740)
    draw = dd.uniform((10, 4), 741, -2.5, 2.5)
    real_normal = torch.normal
    torch.normal = lambda mean, std, **k: mean + std * T(draw)
    try:
        out["noise_a"] = a; out["noise_draw"] = draw
        out["noise_tgt"] = R.add_normal_noise(T(a), std=0.8, noise_bounds=[-0.2, 0.2], out_bounds=[-1., 1.]).numpy()
        out["noise_mixed"] = R.add_mixed_normal_noise(T(a), std_max=0.8, std_min=0.05, out_bounds=[-1., 1.]).numpy()
    finally:
        torch.normal = real_normal
    # RNG consumption: torch.normal(zeros, full(std)) == empty.normal_() * std on one seeded stream
    torch.manual_seed(123)
    ref_draw = torch.normal(torch.zeros(5, 3), torch.full((5, 3), 0.8))
    out["rng_normal_seed123"] = ref_draw.numpy()
    torch.manual_seed(123)
    out["rng_randint_seed123"] = torch.randint(1000, size=(8,)).numpy()


# --------------------------------------------------------------------------- learner traces
def make_cfg(distl=False, B=64, nstep=3):
    return NS(artifact=None,
              algo=NS(batch_size=B, obs_norm=True, distl=distl, gamma=0.99, nstep=nstep, v_min=-10, v_max=10,
                      num_atoms=51, tau=0.05, max_grad_norm=0.5, critic_lr=5e-4, actor_lr=5e-4, memory_size=400,
                      noise=NS(tgt_pol_std=0.8, tgt_pol_noise_bound=0.2)))


class _Capture:
    """Replace torch.randint / torch.normal by deterministic draws and log them."""

    def __init__(self, seed):
        self.seed = seed; self.n = 0; self.idx = []; self.noise = []

    def __enter__(self):
        self._ri, self._no = torch.randint, torch.normal

        def randint(high, size=None, **k):
            self.n += 1
            v = dd.integers(tuple(size), self.seed + self.n, int(high))
            self.idx.append(v)
            return T(v)

        def normal(mean, std, **k):
            self.n += 1
            d = dd.uniform(tuple(mean.shape), self.seed + self.n, -2.0, 2.0)  # stands in for N(0,1) draws
            self.noise.append(d)
            return mean + std * T(d)

        torch.randint, torch.normal = randint, normal
        return self

    def __exit__(self, *a):
        torch.randint, torch.normal = self._ri, self._no


def _fill_data(O, A, rows, seed):
    return (dd.uniform((rows, O), seed, -3, 3), dd.uniform((rows, A), seed + 1), dd.uniform((rows, 1), seed + 2, -0.05, 0.05),
            dd.uniform((rows, O), seed + 3, -3, 3), dd.bernoulli((rows, 1), seed + 4, 0.1))


def gen_learners(R, out, steps=3):
    import torch.nn.functional as F
    O, A = 8, 2
    norm = (dd.uniform((O,), 801, -0.5, 0.5), dd.uniform((O,), 802, 0.5, 2.0))
    for distl in (False, True):
        tag = "vd" if distl else "v"
        cfg = make_cfg(distl)
        v = R.PQLVLearner.__new__(R.PQLVLearner)
        v.cfg, v.obs_dim, v.action_dim, v.device = cfg, (O,), A, torch.device("cpu")
        if distl:
            v.critic = R.DistributionalDoubleQ((O,), A, v_min=-10, v_max=10, num_atoms=51, device="cpu")
            load_state(v.critic, dd.doubleq_state(O, A, 51, 31)); v.loss_fnc = F.binary_cross_entropy
        else:
            v.critic = R.DoubleQ((O,), A)
            load_state(v.critic, dd.doubleq_state(O, A, 1, 21)); v.loss_fnc = F.mse_loss
        v.critic_optimizer = torch.optim.AdamW(v.critic.parameters(), cfg.algo.critic_lr)
        v.critic_target = deepcopy(v.critic)
        v.actor = None
        v.memory = R.ReplayBuffer(capacity=400, obs_dim=(O,), action_dim=A, device="cpu")
        v.loss_tracker = R.Tracker(5); v.update_count = 0; v.normalize_tuple = None; v.sleep_time = 0
        actor = R.TanhMLPPolicy((O,), A); load_state(actor, dd.mlp_state(O, A, 11))
        data = _fill_data(O, A, 300, 810)
        v.update(actor, tuple(T(d) for d in data), (T(norm[0]), T(norm[1]), 1e-4), 0)
        losses = []
        with _Capture(8200 if distl else 8100) as cap:
            for s in range(steps):
                v.learn()
                losses.append(v.loss_tracker.moving_average[-1])
                for k, p in v.critic.named_parameters():
                    out[f"{tag}_s{s}_p_{k}"] = dd.summarize(p.detach().numpy())
                for k, p in v.critic_target.named_parameters():
                    out[f"{tag}_s{s}_t_{k}"] = dd.summarize(p.detach().numpy())
        out[f"{tag}_loss"] = np.array(losses, np.float64)
        out[f"{tag}_idx"] = np.stack(cap.idx); out[f"{tag}_noise"] = np.stack(cap.noise)
        st = v.critic_optimizer.state_dict()["state"]
        out[f"{tag}_adam_step"] = np.array(float(st[0]["step"]))
        out[f"{tag}_adam_m0"] = dd.summarize(st[0]["exp_avg"].numpy()); out[f"{tag}_adam_v0"] = dd.summarize(st[0]["exp_avg_sq"].numpy())
        # final full small tensors for a tight check
        sd = v.critic.state_dict()
        out[f"{tag}_final_q1_last_w"] = sd["net_q1.net.6.weight"].numpy().copy()
        out[f"{tag}_final_q1_last_b"] = sd["net_q1.net.6.bias"].numpy().copy()

    # P-learner
    cfg = make_cfg(False)
    p = R.PQLPLearner.__new__(R.PQLPLearner)
    p.cfg, p.obs_dim, p.action_dim, p.device = cfg, (O,), A, torch.device("cpu")
    p.actor = R.TanhMLPPolicy((O,), A); load_state(p.actor, dd.mlp_state(O, A, 11))
    p.actor_optimizer = torch.optim.AdamW(p.actor.parameters(), cfg.algo.actor_lr)
    p.critic = None; p.memory_size = 400; p.memory = torch.empty((400, O)); p.next_p = 0; p.if_full = False; p.cur_capacity = 0
    p.loss_tracker = R.Tracker(5); p.update_count = 0; p.normalize_tuple = None; p.sleep_time = 0.01
    critic = R.DoubleQ((O,), A); load_state(critic, dd.doubleq_state(O, A, 1, 21))
    p.update(critic, T(_fill_data(O, A, 300, 810)[0]), (T(norm[0]), T(norm[1]), 1e-4), 0)
    losses = []
    with _Capture(8300) as cap:
        for s in range(steps):
            p.learn(); losses.append(p.loss_tracker.moving_average[-1])
            for k, q in p.actor.named_parameters():
                out[f"p_s{s}_p_{k}"] = dd.summarize(q.detach().numpy())
    out["p_loss"] = np.array(losses, np.float64); out["p_idx"] = np.stack(cap.idx)
    out["p_final_last_w"] = p.actor.state_dict()["net.6.weight"].numpy().copy()
    out["learner_norm_mean"] = norm[0]; out["learner_norm_var"] = norm[1]

    # P-learner through a distributional critic (get_q_min -> (B,))
    p2 = R.PQLPLearner.__new__(R.PQLPLearner)
    p2.cfg, p2.obs_dim, p2.action_dim, p2.device = cfg, (O,), A, torch.device("cpu")
    p2.actor = R.TanhMLPPolicy((O,), A); load_state(p2.actor, dd.mlp_state(O, A, 11))
    p2.actor_optimizer = torch.optim.AdamW(p2.actor.parameters(), cfg.algo.actor_lr)
    p2.critic = None; p2.memory_size = 400; p2.memory = torch.empty((400, O)); p2.next_p = 0; p2.if_full = False; p2.cur_capacity = 0
    p2.loss_tracker = R.Tracker(5); p2.update_count = 0; p2.normalize_tuple = None; p2.sleep_time = 0.01
    dcritic = R.DistributionalDoubleQ((O,), A, v_min=-10, v_max=10, num_atoms=51, device="cpu")
    load_state(dcritic, dd.doubleq_state(O, A, 51, 31))
    p2.update(dcritic, T(_fill_data(O, A, 300, 810)[0]), (T(norm[0]), T(norm[1]), 1e-4), 0)
    losses = []
    with _Capture(8400) as cap:
        for s in range(steps):
            p2.learn(); losses.append(p2.loss_tracker.moving_average[-1])
    out["pd_loss"] = np.array(losses, np.float64); out["pd_idx"] = np.stack(cap.idx)
    out["pd_final_last_w"] = p2.actor.state_dict()["net.6.weight"].numpy().copy()


# --------------------------------------------------------------------------- SAC
class _EpsCapture:
    """Replace the standard-normal draw of torch.distributions.Normal.rsample by deterministic values and log them."""

    def __init__(self, seed):
        self.seed = seed; self.n = 0; self.draws = []

    def __enter__(self):
        import torch.distributions.normal as tdn
        self._mod, self._orig = tdn, tdn._standard_normal

        def std_normal(shape, dtype=None, device=None):
            self.n += 1
            d = dd.uniform(tuple(shape), self.seed + self.n, -2.0, 2.0)
            self.draws.append(d)
            return T(d)

        tdn._standard_normal = std_normal
        return self

    def __exit__(self, *a):
        self._mod._standard_normal = self._orig


def gen_sac(R, out, steps=3):
    from torch import nn
    # known-answer vectors of the squashed-Gaussian head: actions, log-prob and parameter gradients of a scalar of both
    for tag, O, A, B in (("kat_toy", 8, 2, 16), ("kat_allegro", 88, 16, 16)):
        pol = R.TanhDiagGaussianMLPPolicy((O,), A)
        state = dd.mlp_state(O, 2 * A, 61)
        if A > 2:   # push some log_std outputs across the [-5, 5] clamp
            state["net.6.bias"] = state["net.6.bias"].copy()
            state["net.6.bias"][A:] = np.linspace(-6.5, 6.5, A).astype(np.float32)
        load_state(pol, state)
        x = T(dd.uniform((B, O), 62, -2, 2))
        w = T(dd.uniform((A,), 63, -1, 1))
        with _EpsCapture(6400) as cap:
            a, _, logp = pol.get_actions_logprob(x)
        loss = (0.3 * logp - (a * w).sum(-1, keepdim=True)).mean()
        grads = torch.autograd.grad(loss, list(pol.parameters()))
        out[f"{tag}_meta"] = np.array([O, A, B]); out[f"{tag}_eps"] = cap.draws[0]
        out[f"{tag}_act"] = a.detach().numpy(); out[f"{tag}_logp"] = logp.detach().numpy(); out[f"{tag}_loss"] = np.array(float(loss.detach()))
        for (k, _), g in zip(pol.named_parameters(), grads):
            out[f"{tag}_g_{k}"] = dd.summarize(g.numpy())
        out[f"{tag}_g_last_b"] = grads[-1].numpy().copy()
        with torch.no_grad():
            out[f"{tag}_mean_act"] = pol.get_actions(x, sample=False).numpy()

    # update_net trace (sac.py:98-108): critic step, actor step, temperature step, Polyak; RNG = idx, eps_next, eps_cur per step
    O, A, B = 8, 2, 64
    norm = (dd.uniform((O,), 801, -0.5, 0.5), dd.uniform((O,), 802, 0.5, 2.0))
    cfg = NS(info_track_keys=None, device="cpu",
             algo=NS(batch_size=B, obs_norm=True, gamma=0.99, nstep=3, tau=0.05, max_grad_norm=0.5, alpha=None, alpha_lr=0.005,
                     no_tgt_actor=True, update_times=1))
    s = R.AgentSAC.__new__(R.AgentSAC)
    s.cfg, s.obs_dim, s.action_dim, s.device = cfg, (O,), A, torch.device("cpu")
    s.actor = R.TanhDiagGaussianMLPPolicy((O,), A); load_state(s.actor, dd.mlp_state(O, 2 * A, 11))
    s.critic = R.DoubleQ((O,), A); load_state(s.critic, dd.doubleq_state(O, A, 1, 21))
    s.critic_target = deepcopy(s.critic); s.actor_target = s.actor
    s.actor_optimizer = torch.optim.AdamW(s.actor.parameters(), 5e-4)
    s.critic_optimizer = torch.optim.AdamW(s.critic.parameters(), 5e-4)
    s.log_alpha = nn.Parameter(torch.zeros(1)); s.alpha_optim = torch.optim.AdamW([s.log_alpha], lr=cfg.algo.alpha_lr)
    s.target_entropy = -A
    s.obs_rms = R.RunningMeanStd(shape=(O,), device="cpu"); s.obs_rms.mean, s.obs_rms.var = T(norm[0]), T(norm[1])
    data = [T(d) for d in _fill_data(O, A, 300, 810)]
    closs, aloss, alphas, idxs = [], [], [], []
    with _EpsCapture(8500) as cap:
        for st in range(steps):
            idx = T(dd.integers((B,), 8600 + st, 300)); idxs.append(idx.numpy())
            obs, act, rew, nobs, done = (d[idx] for d in data)
            obs, nobs = s.obs_rms.normalize(obs), s.obs_rms.normalize(nobs)
            cl, _ = s.update_critic(obs, act, rew, nobs, done)
            al, _ = s.update_actor(obs)
            R.soft_update(s.critic_target, s.critic, cfg.algo.tau)
            closs.append(cl); aloss.append(al); alphas.append(float(s.log_alpha.detach()))
            for k, p in s.actor.named_parameters():
                out[f"sac_s{st}_a_{k}"] = dd.summarize(p.detach().numpy())
            for k, p in s.critic.named_parameters():
                out[f"sac_s{st}_c_{k}"] = dd.summarize(p.detach().numpy())
            for k, p in s.critic_target.named_parameters():
                out[f"sac_s{st}_t_{k}"] = dd.summarize(p.detach().numpy())
    out["sac_closs"] = np.array(closs, np.float64); out["sac_aloss"] = np.array(aloss, np.float64)
    out["sac_log_alpha"] = np.array(alphas, np.float64)
    out["sac_idx"] = np.stack(idxs); out["sac_eps"] = np.stack(cap.draws)   # (2*steps, B, A): next, cur, next, cur, ...
    out["sac_final_actor_last_w"] = s.actor.state_dict()["net.6.weight"].numpy().copy()
    out["sac_final_q1_last_w"] = s.critic.state_dict()["net_q1.net.6.weight"].numpy().copy()
    out["sac_norm_mean"] = norm[0]; out["sac_norm_var"] = norm[1]


# --------------------------------------------------------------------------- CrossQ
def gen_crossq(R, out, steps=3):
    O, A, B = 8, 2, 64
    norm = (dd.uniform((O,), 801, -0.5, 0.5), dd.uniform((O,), 802, 0.5, 2.0))
    cfg = NS(info_track_keys=None, device="cpu",
             algo=NS(batch_size=B, obs_norm=True, gamma=0.99, nstep=3, tau=0.05, max_grad_norm=0.5, no_tgt_actor=True, update_times=1,
                     noise=NS(tgt_pol_std=0.8, tgt_pol_noise_bound=0.2)))
    s = R.AgentCrossQ.__new__(R.AgentCrossQ)
    s.cfg, s.obs_dim, s.action_dim, s.device = cfg, (O,), A, torch.device("cpu")
    s.actor = R.TanhMLPPolicy((O,), A); load_state(s.actor, dd.mlp_state(O, A, 11))
    s.actor_target = s.actor
    s.critic = R.DoubleQBatchNorm((O,), A)
    s.critic.load_state_dict({k: T(v) for k, v in dd.bn_critic_state(O, A, 41).items()}, strict=False)
    s.actor_optimizer = torch.optim.AdamW(s.actor.parameters(), 5e-4)
    s.critic_optimizer = torch.optim.AdamW(s.critic.parameters(), 5e-4)
    s.obs_rms = R.RunningMeanStd(shape=(O,), device="cpu"); s.obs_rms.mean, s.obs_rms.var = T(norm[0]), T(norm[1])
    data = [T(d) for d in _fill_data(O, A, 300, 810)]
    # forward known-answer vector (training mode on a 2B batch, then eval mode with the moved running statistics)
    xk, ak = T(dd.uniform((2 * B, O), 71, -2, 2)), T(dd.uniform((2 * B, A), 72, -1, 1))
    probe = deepcopy(s.critic)
    with torch.no_grad():
        q1, q2 = probe.get_q1_q2(xk, ak)
        out["cq_kat_train_q1"], out["cq_kat_train_q2"] = q1.numpy(), q2.numpy()
        probe.eval()
        q1, q2 = probe.get_q1_q2(xk, ak)
        out["cq_kat_eval_q1"], out["cq_kat_eval_q2"] = q1.numpy(), q2.numpy()
    out["cq_kat_running_mean_l0"] = probe.state_dict()["net_q1.net.1.running_mean"].numpy().copy()
    out["cq_kat_running_var_l0"] = probe.state_dict()["net_q1.net.1.running_var"].numpy().copy()
    closs, aloss, idxs = [], [], []
    with _Capture(8700) as cap:     # torch.normal of add_normal_noise
        for st in range(steps):
            idx = T(dd.integers((B,), 8800 + st, 300)); idxs.append(idx.numpy())
            obs, act, rew, nobs, done = (d[idx] for d in data)
            obs, nobs = s.obs_rms.normalize(obs), s.obs_rms.normalize(nobs)
            cl, _ = s.update_critic(obs, act, rew, nobs, done)
            al, _ = s.update_actor(obs)
            closs.append(cl); aloss.append(al)
            for k, p in s.actor.named_parameters():
                out[f"cq_s{st}_a_{k}"] = dd.summarize(p.detach().numpy())
            for k, p in s.critic.named_parameters():
                out[f"cq_s{st}_c_{k}"] = dd.summarize(p.detach().numpy())
            sd = s.critic.state_dict()
            for k in sd:
                if "running" in k:
                    out[f"cq_s{st}_r_{k}"] = sd[k].numpy().copy()
    out["cq_closs"] = np.array(closs, np.float64); out["cq_aloss"] = np.array(aloss, np.float64)
    out["cq_idx"] = np.stack(idxs); out["cq_noise"] = np.stack(cap.noise)
    out["cq_final_actor_last_w"] = s.actor.state_dict()["net.6.weight"].numpy().copy()
    out["cq_final_q1_last_w"] = s.critic.state_dict()["net_q1.net.9.weight"].numpy().copy()
    out["cq_final_q1_bn0_gamma"] = s.critic.state_dict()["net_q1.net.1.weight"].numpy().copy()
    out["cq_norm_mean"] = norm[0]; out["cq_norm_var"] = norm[1]


# --------------------------------------------------------------------------- DDPG (BASELINE cfg #1)
def gen_ddpg(R, out, steps=3):
    """AgentDDPG.update_net's inner iteration (pql/algo/ddpg.py:119-166) on the reference itself: shared batch, obs_rms.normalize
    (no clamp), update_critic, update_actor, soft_update of the critic target and -- `no_tgt_actor=False` -- of the actor target.
    Built with __new__ + attributes like the SAC / CrossQ traces (its __post_init__ wants an env); draws captured.  Two traces:
    `tgt0` (no_tgt_actor=True: the target actor IS the actor, ddpg.py:22) and `tgt1` (a Polyak-averaged copy, ddpg.py:134-135)."""
    O, A, B = 8, 2, 64
    norm = (dd.uniform((O,), 801, -0.5, 0.5), dd.uniform((O,), 802, 0.5, 2.0))
    data = [T(d) for d in _fill_data(O, A, 300, 810)]
    for tag, no_tgt in (("tgt0", True), ("tgt1", False)):
        cfg = NS(info_track_keys=None, device="cpu",
                 algo=NS(batch_size=B, obs_norm=True, gamma=0.99, nstep=3, tau=0.05, max_grad_norm=0.5, no_tgt_actor=no_tgt, update_times=1,
                         noise=NS(tgt_pol_std=0.8, tgt_pol_noise_bound=0.2)))
        s = R.AgentDDPG.__new__(R.AgentDDPG)
        s.cfg, s.obs_dim, s.action_dim, s.device = cfg, (O,), A, torch.device("cpu")
        s.actor = R.TanhMLPPolicy((O,), A); load_state(s.actor, dd.mlp_state(O, A, 11))
        s.critic = R.DoubleQ((O,), A); load_state(s.critic, dd.doubleq_state(O, A, 1, 21))
        s.critic_target = deepcopy(s.critic)
        s.actor_target = s.actor if no_tgt else deepcopy(s.actor)     # ddpg.py:21-22
        if not no_tgt:   # start the target actor AWAY from the actor, so that a build that reads the wrong one cannot pass
            load_state(s.actor_target, dd.mlp_state(O, A, 13))
        s.actor_optimizer = torch.optim.AdamW(s.actor.parameters(), 5e-4)
        s.critic_optimizer = torch.optim.AdamW(s.critic.parameters(), 5e-4)
        s.obs_rms = R.RunningMeanStd(shape=(O,), device="cpu"); s.obs_rms.mean, s.obs_rms.var = T(norm[0]), T(norm[1])
        closs, aloss, idxs = [], [], []
        with _Capture(9100) as cap:     # torch.normal of add_normal_noise (get_tgt_policy_actions, ddpg.py:70-79)
            for st in range(steps):
                idx = T(dd.integers((B,), 9200 + st, 300)); idxs.append(idx.numpy())
                obs, act, rew, nobs, done = (d[idx] for d in data)
                obs, nobs = s.obs_rms.normalize(obs), s.obs_rms.normalize(nobs)
                cl, _ = s.update_critic(obs, act, rew, nobs, done)
                al, _ = s.update_actor(obs)
                R.soft_update(s.critic_target, s.critic, cfg.algo.tau)
                if not cfg.algo.no_tgt_actor:
                    R.soft_update(s.actor_target, s.actor, cfg.algo.tau)
                closs.append(cl); aloss.append(al)
                for k, p in s.actor.named_parameters():
                    out[f"ddpg_{tag}_s{st}_a_{k}"] = dd.summarize(p.detach().numpy())
                for k, p in s.critic.named_parameters():
                    out[f"ddpg_{tag}_s{st}_c_{k}"] = dd.summarize(p.detach().numpy())
                for k, p in s.critic_target.named_parameters():
                    out[f"ddpg_{tag}_s{st}_t_{k}"] = dd.summarize(p.detach().numpy())
                if not no_tgt:
                    for k, p in s.actor_target.named_parameters():
                        out[f"ddpg_{tag}_s{st}_at_{k}"] = dd.summarize(p.detach().numpy())
        out[f"ddpg_{tag}_closs"] = np.array(closs, np.float64); out[f"ddpg_{tag}_aloss"] = np.array(aloss, np.float64)
        out[f"ddpg_{tag}_idx"] = np.stack(idxs); out[f"ddpg_{tag}_noise"] = np.stack(cap.noise)
        out[f"ddpg_{tag}_final_actor_last_w"] = s.actor.state_dict()["net.6.weight"].numpy().copy()
        out[f"ddpg_{tag}_final_q1_last_w"] = s.critic.state_dict()["net_q1.net.6.weight"].numpy().copy()
        out[f"ddpg_{tag}_final_tq1_last_w"] = s.critic_target.state_dict()["net_q1.net.6.weight"].numpy().copy()
        if not no_tgt:
            out[f"ddpg_{tag}_final_tactor_last_w"] = s.actor_target.state_dict()["net.6.weight"].numpy().copy()
    out["ddpg_norm_mean"] = norm[0]; out["ddpg_norm_var"] = norm[1]


def gen_ckpt(R, out):
    """f1: the ONE weight file the reference ships (pql/model.pth, a PPO actor / critic pair in the checkpoint format of
    pql/utils/model_util.py:24-36) read with weights_only=True and pushed through the reference's MLPNet (mlp.py:27-40):
    the weights themselves (data, 1.6 MB), outputs, input gradients and parameter-gradient fingerprints."""
    ck = torch.load(os.path.join(REF, "pql", "model.pth"), map_location="cpu", weights_only=True)
    assert ck["obs_rms"] is None
    out["actor_logstd"] = ck["actor"]["logstd"].numpy()
    for role, sd, strip in (("actor", ck["actor"], ""), ("critic", ck["critic"], "critic.")):
        sd = {k[len(strip):]: v for k, v in sd.items() if k.startswith(strip + "net.")}
        in_dim, out_dim = sd["net.0.weight"].shape[1], sd["net.6.weight"].shape[0]
        net = R.MLPNet(in_dim, out_dim)     # default hidden [512, 256, 128] = the file's
        net.load_state_dict(sd)
        B = 19
        x = T(dd.uniform((B, in_dim), 7000 + out_dim, -2, 2)).requires_grad_(True)
        y = net(x)
        w = T(dd.uniform((B, out_dim), 7100 + out_dim))
        (y * w).sum().backward()
        for k, v in sd.items():
            out[f"{role}_w_{k}"] = v.numpy()
        out[f"{role}_y"] = y.detach().numpy()
        out[f"{role}_dx"] = x.grad.numpy()
        for k, p in net.named_parameters():
            out[f"{role}_g_{k}"] = dd.summarize(p.grad.numpy())


def main():
    torch.set_num_threads(1)
    torch.manual_seed(0)
    R = _import_reference()
    os.makedirs(OUT, exist_ok=True)
    only = set(sys.argv[1:])
    for name, fn in (("replay", gen_ring), ("nstep", gen_nstep), ("models", gen_models), ("math", gen_math),
                     ("learners", gen_learners), ("sac", gen_sac), ("crossq", gen_crossq), ("ddpg", gen_ddpg), ("ckpt", gen_ckpt)):
        if only and name not in only:
            continue
        out = {}
        with torch.no_grad() if name in ("replay", "nstep") else contextlib.nullcontext():
            fn(R, out)
        path = os.path.join(OUT, f"{name}.npz")
        np.savez_compressed(path, **out)
        print(f"{name}: {len(out)} arrays -> {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


if __name__ == "__main__":
    main()
