"""Pin the CPU oracle (oracle/pql_ref_cpu.py) to the golden vectors the reference produced
(tools/gen_golden.py).  CPU only.  Byte/index work is compared bit-exactly; floating point at the
tolerance written next to each assert."""
import numpy as np
import pytest
import torch

import detdata as dd
from oracle import pql_ref_cpu as ref

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731


# --------------------------------------------------------------------------- ring
@pytest.mark.parametrize("name", ["wrap4", "exact", "ragged"])
def test_ring_pointer_trace_and_contents(golden, name):
    g = golden("replay")
    cap, O, A = g[f"ring_{name}_meta"]
    ring = ref.RingRef(cap, O, A)
    for step, m in enumerate(g[f"ring_{name}_inserts"]):
        seed = 100 + step
        ring.insert(T(dd.uniform((m, O), seed)), T(dd.uniform((m, A), seed + 20)), T(dd.uniform((m, 1), seed + 40)),
                    T(dd.uniform((m, O), seed + 60)), T(dd.bernoulli((m, 1), seed + 80, 0.3)))
        assert [ring.next_p, ring.cur_capacity, int(ring.if_full)] == g[f"ring_{name}_trace"][step].tolist()
    for nm, t in (("obs", ring.obs), ("act", ring.act), ("rew", ring.rew), ("nobs", ring.nobs), ("done", ring.done)):
        assert np.array_equal(t.numpy(), g[f"ring_{name}_{nm}"]), nm   # bit-exact
    out = ring.gather(T(g[f"ring_{name}_idx"]))
    for nm, t in zip(("s_obs", "s_act", "s_rew", "s_nobs", "s_done"), out):
        assert t.dtype == torch.float32
        assert np.array_equal(t.numpy(), g[f"ring_{name}_{nm}"]), nm


def test_obs_ring(golden):
    g = golden("replay")
    r = ref.ObsRingRef(10, 4)
    for step, m in enumerate([4, 4, 4, 4, 10]):
        r.insert(T(dd.uniform((m, 4), 300 + step)))
        assert [r.next_p, r.cur_capacity, int(r.if_full)] == g["pring_trace"][step].tolist()
    assert np.array_equal(r.mem.numpy(), g["pring_mem"])


def test_ring_plan_rejects_oversize():
    with pytest.raises(ValueError):
        ref.ring_plan(3, False, 10, 25)


# --------------------------------------------------------------------------- n-step
@pytest.mark.parametrize("name", ["kat3", "n3", "n5", "n1"])
def test_nstep(golden, name):
    g = golden("nstep")
    meta = g[f"nstep_{name}_meta"]
    N, n, O, A = meta[:4]
    calls = meta[4:]
    ns = ref.NStepRef(O, A, N, n)
    assert np.array_equal(ns.gamma_pow.numpy().reshape(-1, 1), g[f"nstep_{name}_gamma"])
    for ci, Tn in enumerate(calls):
        seed = 500 + 10 * ci
        obs = dd.uniform((N, Tn, O), seed); act = dd.uniform((N, Tn, A), seed + 1)
        rew = dd.uniform((N, Tn, 1), seed + 2); nobs = dd.uniform((N, Tn, O), seed + 3)
        done = dd.bernoulli((N, Tn, 1), seed + 4, 0.25)
        if name == "kat3":
            rew, done = g["nstep_kat3_in_rew"], g["nstep_kat3_in_done"]
        res = ns.add(T(obs), T(act), T(rew), T(nobs), T(done))
        for nm, t in zip(("obs", "act", "rew", "nobs", "done"), res):
            exp = g[f"nstep_{name}_c{ci}_{nm}"]
            assert tuple(t.shape) == exp.shape, (nm, ci)
            assert np.array_equal(t.numpy(), exp), (nm, ci)   # bit-exact, including the n-term fp32 reward sum


def test_nstep_kat_values(golden):
    """SURVEY section 4 known answers: R = 2.9701 / 1.99 / 1.0 and time-major row order."""
    g = golden("nstep")
    r = g["nstep_kat3_c0_rew"].ravel()
    np.testing.assert_allclose(r, [2.9701, 1.99, 2.9701, 2.9701, 1.0, 2.9701, 2.9701, 2.9701, 2.9701], atol=1e-6)
    assert g["nstep_kat3_c0_done"].ravel().tolist() == [0, 1, 0, 0, 1, 0, 0, 0, 1]


def test_nstep_first_call_shorter_than_window_raises():
    ns = ref.NStepRef(2, 1, 3, 3)
    z = torch.zeros
    with pytest.raises((RuntimeError, ValueError)):   # reference: torch.cat([]) at nstep_replay.py:65
        ns.add(z(3, 2, 2), z(3, 2, 1), z(3, 2, 1), z(3, 2, 2), z(3, 2, 1))


# --------------------------------------------------------------------------- models
SHAPES = [(8, 2), (88, 16), (211, 20), (108, 21)]


def _grads_close(named_params, g, prefix, rtol=2e-4, atol=2e-6):
    for k, p in named_params:
        exp = g[f"{prefix}{k}"]
        got = dd.summarize(p.numpy())
        np.testing.assert_allclose(got, exp, rtol=rtol, atol=atol, err_msg=k)


def _named(prefix_list, plist):
    names = []
    for pre, params in zip(prefix_list, plist):
        for i in range(len(params) // 2):
            names += [f"{pre}{2 * i}.weight", f"{pre}{2 * i}.bias"]
    return names


@pytest.mark.parametrize("O,A", SHAPES)
def test_actor_forward_backward(golden, O, A):
    g = golden("models"); tag = f"o{O}a{A}"; B = 33 if O == 8 else 17
    obs = T(dd.uniform((B, O), 1000 + O, -2, 2)).requires_grad_(True)
    params = [p.requires_grad_(True) for p in ref.params_from_state(dd.mlp_state(O, A, 11))]
    y = ref.actor_forward_ref(params, obs)
    np.testing.assert_allclose(y.detach().numpy(), g[f"actor_{tag}_y"], atol=1e-6)   # fp32 forward: 1e-6
    gr = torch.autograd.grad((y * T(dd.uniform((B, A), 3000 + O))).sum(), [obs, *params])
    np.testing.assert_allclose(gr[0].numpy(), g[f"actor_{tag}_dobs"], atol=1e-6)
    _grads_close(zip(_named(["net."], [params]), gr[1:]), g, f"actor_{tag}_g_")


@pytest.mark.parametrize("O,A", SHAPES)
def test_doubleq_forward_backward(golden, O, A):
    g = golden("models"); tag = f"o{O}a{A}"; B = 33 if O == 8 else 17
    obs = T(dd.uniform((B, O), 1000 + O, -2, 2)).requires_grad_(True); act = T(dd.uniform((B, A), 2000 + O)).requires_grad_(True)
    st = dd.doubleq_state(O, A, 1, 21)
    q1 = [p.requires_grad_(True) for p in ref.params_from_state(st, "net_q1.net.")]
    q2 = [p.requires_grad_(True) for p in ref.params_from_state(st, "net_q2.net.")]
    a, b = ref.twin_forward_ref(q1, q2, obs, act)
    np.testing.assert_allclose(a.detach().numpy(), g[f"dq_{tag}_q1"], atol=1e-5)   # north_star: Q within 1e-5
    np.testing.assert_allclose(b.detach().numpy(), g[f"dq_{tag}_q2"], atol=1e-5)
    np.testing.assert_allclose(ref.qmin_ref(q1, q2, obs, act).detach().numpy(), g[f"dq_{tag}_qmin"], atol=1e-5)
    tgt = T(dd.uniform((B, 1), 4000 + O))
    loss = torch.nn.functional.mse_loss(a, tgt) + torch.nn.functional.mse_loss(b, tgt)
    np.testing.assert_allclose(loss.item(), g[f"dq_{tag}_loss"], rtol=1e-6)
    gr = torch.autograd.grad(loss, [obs, act, *q1, *q2])
    np.testing.assert_allclose(gr[0].numpy(), g[f"dq_{tag}_dobs"], atol=1e-6)
    np.testing.assert_allclose(gr[1].numpy(), g[f"dq_{tag}_dact"], atol=1e-6)
    _grads_close(zip(_named(["net_q1.net.", "net_q2.net."], [q1, q2]), gr[2:]), g, f"dq_{tag}_g_")
    o2 = obs.detach().clone().requires_grad_(True); a2 = act.detach().clone().requires_grad_(True)
    gd = torch.autograd.grad(-ref.qmin_ref(q1, q2, o2, a2).mean(), [a2, o2])
    np.testing.assert_allclose(gd[0].numpy(), g[f"dq_{tag}_dpg_dact"], atol=1e-7)
    np.testing.assert_allclose(gd[1].numpy(), g[f"dq_{tag}_dpg_dobs"], atol=1e-7)


@pytest.mark.parametrize("O,A", SHAPES)
def test_distributional_forward_backward(golden, O, A):
    g = golden("models"); tag = f"o{O}a{A}"; B = 33 if O == 8 else 17; K = 51
    obs = T(dd.uniform((B, O), 1000 + O, -2, 2)).requires_grad_(True); act = T(dd.uniform((B, A), 2000 + O)).requires_grad_(True)
    st = dd.doubleq_state(O, A, K, 31)
    q1 = [p.requires_grad_(True) for p in ref.params_from_state(st, "net_q1.net.")]
    q2 = [p.requires_grad_(True) for p in ref.params_from_state(st, "net_q2.net.")]
    z = torch.linspace(-10, 10, K)
    assert np.array_equal(z.numpy(), g[f"ddq_{tag}_z"])
    p1, p2 = ref.twin_dist_ref(q1, q2, obs, act)
    np.testing.assert_allclose(p1.detach().numpy(), g[f"ddq_{tag}_p1"], atol=1e-6)
    np.testing.assert_allclose(p2.detach().numpy(), g[f"ddq_{tag}_p2"], atol=1e-6)
    np.testing.assert_allclose(ref.qmin_ref(q1, q2, obs, act, z).detach().numpy(), g[f"ddq_{tag}_qmin"], atol=1e-5)
    tg = T(g[f"ddq_{tag}_tgt"])
    loss = torch.nn.functional.binary_cross_entropy(p1, tg) + torch.nn.functional.binary_cross_entropy(p2, tg)
    np.testing.assert_allclose(loss.item(), g[f"ddq_{tag}_loss"], rtol=1e-6)
    gr = torch.autograd.grad(loss, [obs, act, *q1, *q2])
    np.testing.assert_allclose(gr[0].numpy(), g[f"ddq_{tag}_dobs"], atol=1e-6)
    _grads_close(zip(_named(["net_q1.net.", "net_q2.net."], [q1, q2]), gr[2:]), g, f"ddq_{tag}_g_")


def test_baseline_hidden_shape(golden):
    g = golden("models"); hid = (512, 512, 256)
    params = ref.params_from_state(dd.mlp_state(104, 1, 41, hid))
    x = T(dd.uniform((9, 104), 6000)).requires_grad_(True)
    y = ref.mlp_forward_ref(params, x)
    np.testing.assert_allclose(y.detach().numpy(), g["mlp_h512_512_256_y"], atol=1e-6)
    np.testing.assert_allclose(torch.autograd.grad(y.sum(), x)[0].numpy(), g["mlp_h512_512_256_dx"], atol=1e-6)


def test_reference_checkpoint_weights_through_the_oracle(golden):
    """f1: the reference's own pql/model.pth (weights stored as arrays in the fixture) through the oracle's MLP == the
    reference MLPNet's outputs and gradients on it."""
    g = golden("ckpt")
    for role, out_dim in (("actor", 12), ("critic", 1)):
        st = {k[len(role) + 3:]: g[k] for k in g if k.startswith(f"{role}_w_")}
        params = [p.requires_grad_(True) for p in ref.params_from_state(st)]
        x = T(dd.uniform((19, 63), 7000 + out_dim, -2, 2)).requires_grad_(True)
        y = ref.mlp_forward_ref(params, x)
        np.testing.assert_allclose(y.detach().numpy(), g[f"{role}_y"], atol=1e-6)
        w = T(dd.uniform((19, out_dim), 7100 + out_dim))
        gr = torch.autograd.grad((y * w).sum(), [x, *params])
        np.testing.assert_allclose(gr[0].numpy(), g[f"{role}_dx"], atol=1e-6)
        _grads_close(zip(_named(["net."], [params]), gr[1:]), g, f"{role}_g_")


# --------------------------------------------------------------------------- math
def test_projection(golden):
    g = golden("math")
    out = ref.c51_project_ref(T(g["proj_p"]), T(g["proj_rew"]), T(g["proj_done"]), float(g["proj_gamma"]), -10, 10, 51)
    np.testing.assert_allclose(out.numpy(), g["proj_out"], atol=1e-7)     # C51: compare at <=1e-6 (SURVEY hard parts)
    assert np.array_equal(out.numpy() != 0, g["proj_out"] != 0)           # same support bins, incl. integral-b edge rows
    out2 = ref.c51_project_ref(T(g["proj2_p"]), T(g["proj2_rew"]), T(g["proj2_done"]), 0.95, -2, 6, 11)
    np.testing.assert_allclose(out2.numpy(), g["proj2_out"], atol=1e-7)
    # SURVEY section 4 KATs
    assert np.nonzero(g["proj_out"][0])[0].tolist() == [17, 18]
    assert np.nonzero(g["proj_out"][1])[0].tolist() == [50]


def test_normalize_rms_noise(golden):
    g = golden("math")
    y = ref.normalize_ref(T(g["norm_x"]), (T(g["norm_mean"]), T(g["norm_var"]), 1e-4))
    assert np.array_equal(y.numpy(), g["norm_y"])
    rms = ref.RunningMeanStdRef((6,))
    for i in range(3):
        rms.update(T(dd.uniform((32, 6), 730 + i, -3, 5)))
        st = np.concatenate([rms.mean.numpy(), rms.var.numpy(), [rms.count]])
        np.testing.assert_allclose(st, g["rms_trace"][i], rtol=1e-6)
    np.testing.assert_allclose(ref.normalize_ref(T(g["norm_x"]), rms.states(), clamp=False).numpy(), g["rms_norm"], rtol=1e-6)
    a, d = T(g["noise_a"]), T(g["noise_draw"])
    assert np.array_equal(ref.target_noise_ref(a, d, 0.8, 0.2).numpy(), g["noise_tgt"])
    np.testing.assert_allclose(ref.mixed_noise_ref(a, d, 0.05, 0.8).numpy(), g["noise_mixed"], atol=1e-7)


def test_rng_draw_equivalence(golden):
    """torch.normal(zeros, full(std)) consumes the generator exactly like empty.normal_() then *std
    (SURVEY Appendix B): the product path draws that way, so seeded streams line up."""
    g = golden("math")
    torch.manual_seed(123)
    d = torch.empty(5, 3).normal_() * 0.8
    assert np.array_equal(d.numpy(), g["rng_normal_seed123"])
    torch.manual_seed(123)
    assert np.array_equal(torch.randint(1000, size=(8,)).numpy(), g["rng_randint_seed123"])


# --------------------------------------------------------------------------- learners
def _fill(O, A, rows, seed):
    return (T(dd.uniform((rows, O), seed, -3, 3)), T(dd.uniform((rows, A), seed + 1)), T(dd.uniform((rows, 1), seed + 2, -0.05, 0.05)),
            T(dd.uniform((rows, O), seed + 3, -3, 3)), T(dd.bernoulli((rows, 1), seed + 4, 0.1)))


def _check_params(named, g, prefix, rtol=5e-5, atol=5e-7):
    for k, p in named:
        np.testing.assert_allclose(dd.summarize(p.detach().numpy()), g[f"{prefix}{k}"], rtol=rtol, atol=atol, err_msg=prefix + k)


@pytest.mark.parametrize("distl", [False, True])
def test_v_learner_trace(golden, distl):
    g = golden("learners"); tag = "vd" if distl else "v"; O, A = 8, 2
    hp = ref.HyperRef(batch_size=64, distl=distl)
    st = dd.doubleq_state(O, A, 51 if distl else 1, 31 if distl else 21)
    v = ref.VLearnerRef(O, A, hp, 400, ref.params_from_state(st, "net_q1.net."), ref.params_from_state(st, "net_q2.net."))
    norm = (T(g["learner_norm_mean"]), T(g["learner_norm_var"]), 1e-4)
    v.update(ref.params_from_state(dd.mlp_state(O, A, 11)), _fill(O, A, 300, 810), norm)
    names = _named(["net_q1.net.", "net_q2.net."], [v.q1, v.q2])
    for s in range(3):
        loss = v.learn(idx=T(g[f"{tag}_idx"][s]), draw=T(g[f"{tag}_noise"][s]))
        np.testing.assert_allclose(loss, g[f"{tag}_loss"][s], rtol=2e-5)
        _check_params(zip(names, [*v.q1, *v.q2]), g, f"{tag}_s{s}_p_")
        _check_params(zip(names, [*v.t1, *v.t2]), g, f"{tag}_s{s}_t_")
    np.testing.assert_allclose(v.q1[-2].detach().numpy(), g[f"{tag}_final_q1_last_w"], rtol=5e-5, atol=5e-7)
    np.testing.assert_allclose(dd.summarize(v.opt.m[0].numpy()), g[f"{tag}_adam_m0"], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(dd.summarize(v.opt.v[0].numpy()), g[f"{tag}_adam_v0"], rtol=1e-4, atol=1e-12)
    assert v.opt.step == int(g[f"{tag}_adam_step"]) == 3


@pytest.mark.parametrize("distl", [False, True])
def test_p_learner_trace(golden, distl):
    g = golden("learners"); tag = "pd" if distl else "p"; O, A = 8, 2
    hp = ref.HyperRef(batch_size=64, distl=distl)
    p = ref.PLearnerRef(O, A, hp, 400, ref.params_from_state(dd.mlp_state(O, A, 11)))
    st = dd.doubleq_state(O, A, 51 if distl else 1, 31 if distl else 21)
    norm = (T(g["learner_norm_mean"]), T(g["learner_norm_var"]), 1e-4)
    p.update(ref.params_from_state(st, "net_q1.net."), ref.params_from_state(st, "net_q2.net."), _fill(O, A, 300, 810)[0], norm)
    for s in range(3):
        loss = p.learn(idx=T(g[f"{tag}_idx"][s]))
        np.testing.assert_allclose(loss, g[f"{tag}_loss"][s], rtol=2e-5)
        if not distl:
            _check_params(zip(_named(["net."], [p.actor]), p.actor), g, f"p_s{s}_p_")
    np.testing.assert_allclose(p.actor[-2].detach().numpy(), g[f"{tag}_final_last_w"], rtol=5e-5, atol=5e-7)


# --------------------------------------------------------------------------- SAC
@pytest.mark.parametrize("tag", ["kat_toy", "kat_allegro"])
def test_squashed_gaussian_head(golden, tag):
    """Oracle head vs the reference's TanhDiagGaussianMLPPolicy.get_actions_logprob with the rsample draw injected;
    kat_allegro pushes log_std across the +-5 clamp and u into tanh saturation."""
    g = golden("sac")
    O, A, B = (int(v) for v in g[f"{tag}_meta"])
    state = dd.mlp_state(O, 2 * A, 61)
    if A > 2:
        state["net.6.bias"] = state["net.6.bias"].copy()
        state["net.6.bias"][A:] = np.linspace(-6.5, 6.5, A).astype(np.float32)
    params = [p.requires_grad_(True) for p in ref.params_from_state(state)]
    x, w = T(dd.uniform((B, O), 62, -2, 2)), T(dd.uniform((A,), 63, -1, 1))
    a, logp = ref.squashed_gaussian_ref(params, x, T(g[f"{tag}_eps"]))
    np.testing.assert_allclose(a.detach().numpy(), g[f"{tag}_act"], atol=1e-6)
    np.testing.assert_allclose(logp.detach().numpy(), g[f"{tag}_logp"], rtol=1e-6, atol=1e-5)
    loss = (0.3 * logp - (a * w).sum(-1, keepdim=True)).mean()
    np.testing.assert_allclose(float(loss.detach()), float(g[f"{tag}_loss"]), rtol=1e-6)
    gr = torch.autograd.grad(loss, params)
    _grads_close(zip(_named(["net."], [params]), gr), g, f"{tag}_g_")
    np.testing.assert_allclose(gr[-1].numpy(), g[f"{tag}_g_last_b"], rtol=2e-4, atol=2e-6)
    with torch.no_grad():
        mu = ref.mlp_forward_ref(params, x)[:, :A]
    np.testing.assert_allclose(mu.tanh().numpy(), g[f"{tag}_mean_act"], atol=1e-6)


def test_sac_trace(golden):
    """SACRef vs three iterations of the reference's AgentSAC.update_critic / update_actor / soft_update."""
    g = golden("sac"); O, A = 8, 2
    hp = ref.HyperRef(batch_size=64)
    st = dd.doubleq_state(O, A, 1, 21)
    s = ref.SACRef(O, A, hp, 400, ref.params_from_state(dd.mlp_state(O, 2 * A, 11)), ref.params_from_state(st, "net_q1.net."),
                   ref.params_from_state(st, "net_q2.net."), alpha_lr=0.005)
    s.ring.insert(*_fill(O, A, 300, 810))
    s.norm = (T(g["sac_norm_mean"]), T(g["sac_norm_var"]), 1e-4)
    cn = _named(["net_q1.net.", "net_q2.net."], [s.q1, s.q2]); an = _named(["net."], [s.actor])
    for i in range(3):
        cl, al, _ = s.update_once(T(g["sac_idx"][i]), T(g["sac_eps"][2 * i]), T(g["sac_eps"][2 * i + 1]))
        np.testing.assert_allclose(cl, g["sac_closs"][i], rtol=2e-5)
        np.testing.assert_allclose(al, g["sac_aloss"][i], rtol=2e-5)
        np.testing.assert_allclose(float(s.log_alpha.detach()), g["sac_log_alpha"][i], rtol=1e-5)
        _check_params(zip(an, s.actor), g, f"sac_s{i}_a_")
        _check_params(zip(cn, [*s.q1, *s.q2]), g, f"sac_s{i}_c_")
        _check_params(zip(cn, [*s.t1, *s.t2]), g, f"sac_s{i}_t_")
    np.testing.assert_allclose(s.actor[-2].detach().numpy(), g["sac_final_actor_last_w"], rtol=5e-5, atol=5e-7)
    np.testing.assert_allclose(s.q1[-2].detach().numpy(), g["sac_final_q1_last_w"], rtol=5e-5, atol=5e-7)


@pytest.mark.parametrize("tag", ["tgt0", "tgt1"])
def test_ddpg_trace(golden, tag):
    """DDPGRef (row a26) vs three iterations of the reference's own AgentDDPG.update_critic / update_actor / soft_update
    (pql/algo/ddpg.py:119-166): tgt0 = no_tgt_actor=True (the shipped configs), tgt1 = a Polyak-averaged target actor that starts
    away from the actor."""
    g = golden("ddpg"); O, A = 8, 2
    st = dd.doubleq_state(O, A, 1, 21)
    tgt = ref.params_from_state(dd.mlp_state(O, A, 13)) if tag == "tgt1" else None
    s = ref.DDPGRef(O, A, ref.HyperRef(batch_size=64), 400, ref.params_from_state(dd.mlp_state(O, A, 11)),
                    ref.params_from_state(st, "net_q1.net."), ref.params_from_state(st, "net_q2.net."), actor_target=tgt)
    s.ring.insert(*_fill(O, A, 300, 810))
    s.norm = (T(g["ddpg_norm_mean"]), T(g["ddpg_norm_var"]), 1e-4)
    cn = _named(["net_q1.net.", "net_q2.net."], [s.q1, s.q2]); an = _named(["net."], [s.actor])
    for i in range(3):
        cl, al = s.update_once(T(g[f"ddpg_{tag}_idx"][i]), T(g[f"ddpg_{tag}_noise"][i]))
        np.testing.assert_allclose(cl, g[f"ddpg_{tag}_closs"][i], rtol=2e-5)
        np.testing.assert_allclose(al, g[f"ddpg_{tag}_aloss"][i], rtol=2e-5)
        _check_params(zip(an, s.actor), g, f"ddpg_{tag}_s{i}_a_")
        _check_params(zip(cn, [*s.q1, *s.q2]), g, f"ddpg_{tag}_s{i}_c_")
        _check_params(zip(cn, [*s.t1, *s.t2]), g, f"ddpg_{tag}_s{i}_t_")
        if tag == "tgt1":
            _check_params(zip(an, s.actor_t), g, f"ddpg_{tag}_s{i}_at_")
    np.testing.assert_allclose(s.actor[-2].detach().numpy(), g[f"ddpg_{tag}_final_actor_last_w"], rtol=5e-5, atol=5e-7)
    np.testing.assert_allclose(s.q1[-2].detach().numpy(), g[f"ddpg_{tag}_final_q1_last_w"], rtol=5e-5, atol=5e-7)
    np.testing.assert_allclose(s.t1[-2].detach().numpy(), g[f"ddpg_{tag}_final_tq1_last_w"], rtol=5e-5, atol=5e-7)
    if tag == "tgt1":
        np.testing.assert_allclose(s.actor_t[-2].numpy(), g[f"ddpg_{tag}_final_tactor_last_w"], rtol=5e-5, atol=5e-7)
        assert not np.allclose(g["ddpg_tgt1_closs"], g["ddpg_tgt0_closs"])   # the two settings really differ


# --------------------------------------------------------------------------- CrossQ
def _bn_lists(state, hidden=(512, 256, 128)):
    lin, bn = [], []
    for pre in ("net_q1.net.", "net_q2.net."):
        lin.append([T(state[f"{pre}{3 * l}.{k}"]) for l in range(len(hidden) + 1) for k in ("weight", "bias")])
        bn.append([T(state[f"{pre}{3 * l + 1}.{k}"]) for l in range(len(hidden)) for k in ("weight", "bias")])
    return lin, bn


def test_crossq_trace(golden):
    """CrossQRef vs the reference's DoubleQBatchNorm (train- and eval-mode forward incl. running statistics) and three
    iterations of AgentCrossQ.update_critic / update_actor."""
    g = golden("crossq"); O, A, B = 8, 2, 64
    lin, bn = _bn_lists(dd.bn_critic_state(O, A, 41))
    s = ref.CrossQRef(O, A, ref.HyperRef(batch_size=B), 400, ref.params_from_state(dd.mlp_state(O, A, 11)), lin, bn)
    # known-answer forward on a probe copy of the statistics
    xk, ak = T(dd.uniform((2 * B, O), 71, -2, 2)), T(dd.uniform((2 * B, A), 72, -1, 1))
    saved = [[t.clone() for t in st] for st in s.q_stats]
    with torch.no_grad():
        q1, q2 = s.q12(xk, ak)
        np.testing.assert_allclose(q1.numpy(), g["cq_kat_train_q1"], atol=2e-6); np.testing.assert_allclose(q2.numpy(), g["cq_kat_train_q2"], atol=2e-6)
        np.testing.assert_allclose(s.q_stats[0][0].numpy(), g["cq_kat_running_mean_l0"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(s.q_stats[0][1].numpy(), g["cq_kat_running_var_l0"], rtol=1e-6)
        x = torch.cat((xk, ak), dim=1)
        e1 = ref.bn_mlp_forward_ref(s.q_lin[0], s.q_bn[0], s.q_stats[0], x, training=False)
        e2 = ref.bn_mlp_forward_ref(s.q_lin[1], s.q_bn[1], s.q_stats[1], x, training=False)
        np.testing.assert_allclose(e1.numpy(), g["cq_kat_eval_q1"], atol=2e-6); np.testing.assert_allclose(e2.numpy(), g["cq_kat_eval_q2"], atol=2e-6)
    s.q_stats = saved
    s.ring.insert(*_fill(O, A, 300, 810))
    s.norm = (T(g["cq_norm_mean"]), T(g["cq_norm_var"]), 1e-4)
    for i in range(3):
        cl, al = s.update_once(T(g["cq_idx"][i]), T(g["cq_noise"][i]))
        np.testing.assert_allclose(cl, g["cq_closs"][i], rtol=2e-5)
        np.testing.assert_allclose(al, g["cq_aloss"][i], rtol=2e-5)
        _check_params(zip(_named(["net."], [s.actor]), s.actor), g, f"cq_s{i}_a_")
        for n, pre in enumerate(("net_q1.net.", "net_q2.net.")):
            for l in range(4):
                for j, k in enumerate(("weight", "bias")):
                    # the bias of a Linear that feeds a BatchNorm has an analytically ZERO gradient (the norm removes any
                    # per-column shift): what reaches AdamW is rounding noise, which m / sqrt(v) turns into +-lr steps of
                    # arbitrary sign -- not comparable between any two implementations, and without effect on the outputs
                    if k == "bias" and l < 3:
                        continue
                    np.testing.assert_allclose(dd.summarize(s.q_lin[n][2 * l + j].detach().numpy()), g[f"cq_s{i}_c_{pre}{3 * l}.{k}"],
                                               rtol=5e-5, atol=1e-5, err_msg=f"{pre}{3 * l}.{k}")
                    if l < 3:
                        np.testing.assert_allclose(dd.summarize(s.q_bn[n][2 * l + j].detach().numpy()), g[f"cq_s{i}_c_{pre}{3 * l + 1}.{k}"],
                                                   rtol=5e-5, atol=1e-5)   # 2 % of one lr step (Adam on ~0 gradients)
                if l < 3:
                    # running_mean contains the (incomparable, see above) pre-norm bias: <= momentum * sum of its +-lr steps
                    np.testing.assert_allclose(s.q_stats[n][2 * l].numpy(), g[f"cq_s{i}_r_{pre}{3 * l + 1}.running_mean"], rtol=2e-5, atol=5e-4)
                    np.testing.assert_allclose(s.q_stats[n][2 * l + 1].numpy(), g[f"cq_s{i}_r_{pre}{3 * l + 1}.running_var"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(s.actor[-2].detach().numpy(), g["cq_final_actor_last_w"], rtol=5e-5, atol=5e-7)
    np.testing.assert_allclose(s.q_lin[0][-2].detach().numpy(), g["cq_final_q1_last_w"], rtol=5e-5, atol=1e-5)
    np.testing.assert_allclose(s.q_bn[0][0].detach().numpy(), g["cq_final_q1_bn0_gamma"], rtol=5e-5, atol=1e-5)
