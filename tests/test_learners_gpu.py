"""Learner-level parity on the GPU: PQLVLearner.learn / PQLPLearner.learn (HIP launch sequences) against
(a) the golden step traces the reference produced (tests/golden/learners.npz: identical replay samples and
noise injected) and (b) the CPU oracle at the BASELINE batch size.  Run with `pytest -m gpu`."""
import numpy as np
import pytest
import torch

import detdata as dd

pytestmark = pytest.mark.gpu

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def make_cfg(distl=False, B=64, memory=400, hidden=None, graph=False, nstep=3, streams=False, task=None):
    """streams=False: learner work on the caller's stream (results readable right after the call); streams=True: the
    learners' own HIP streams with event-fenced hand-offs -- read results after `learner.synchronize()`."""
    from pql_amd.utils.cfg import load_cfg
    ov = [f"algo.batch_size={B}", f"algo.memory_size={memory}", f"algo.distl={distl}", "algo.v_learner_gpu=0",
          "algo.p_learner_gpu=0", "algo.num_gpus=1", f"algo.graph={graph}", f"algo.nstep={nstep}", f"algo.streams={streams}"]
    if task:
        ov.append(f"task={task}")
    cfg = load_cfg(ov)
    cfg.algo.hidden_layers = hidden
    return cfg


def _sd(state):
    return {k: T(v) for k, v in state.items()}


def _fill(O, A, rows, seed):
    return (T(dd.uniform((rows, O), seed, -3, 3)), T(dd.uniform((rows, A), seed + 1)), T(dd.uniform((rows, 1), seed + 2, -0.05, 0.05)),
            T(dd.uniform((rows, O), seed + 3, -3, 3)), T(dd.bernoulli((rows, 1), seed + 4, 0.1)))


def _check_module(module, g, prefix, rtol=5e-5, atol=5e-7):
    """Fingerprint [sum, l2, probes...] of every parameter tensor against the golden trace.  The probes and the l2 norm are held to
    rtol / atol.  The plain SUM of a weight matrix is a cancelling statistic (|sum| ~ 0.1 for an l2 of ~9 over 131 072 weights), and
    AdamW turns the rounding noise of near-zero gradients into updates of up to ~0.1 lr (g ~ eps = 1e-8: lr g / (|g| + eps)), so it is
    held to an absolute bar scaled by the tensor's l2 norm instead: 2e-6 l2 (a 7e-6 drift of that sum after three steps, probes
    intact, is what the DDPG trace shows between torch's summation order and the MFMA k-order)."""
    for key, view in module.named_views():
        got, want = dd.summarize(view.cpu().numpy()), g[f"{prefix}{key}"]
        np.testing.assert_allclose(got[1:], want[1:], rtol=rtol, atol=atol, err_msg=prefix + key)
        np.testing.assert_allclose(got[0], want[0], rtol=rtol, atol=atol + 2e-6 * float(want[1]), err_msg=prefix + key + " (sum)")


@pytest.mark.parametrize("streams", [False, True])
@pytest.mark.parametrize("distl", [False, True])
def test_v_learner_golden_trace(golden, dev, distl, streams):
    from pql_amd.algo.pql_v_learner import PQLVLearner
    from pql_amd.models.mlp import TanhMLPPolicy
    g = golden("learners"); tag = "vd" if distl else "v"; O, A = 8, 2
    v = PQLVLearner((O,), A, make_cfg(distl, streams=streams))
    v.critic.load_state_dict(_sd(dd.doubleq_state(O, A, 51 if distl else 1, 31 if distl else 21)))
    v.critic_target.arena.data.copy_(v.critic.arena.data)
    actor = TanhMLPPolicy((O,), A).to(dev); actor.load_state_dict(_sd(dd.mlp_state(O, A, 11)))
    norm = (T(g["learner_norm_mean"]).to(dev), T(g["learner_norm_var"]).to(dev), 1e-4)
    critic, loss_mean, count = v.update(actor, tuple(t.to(dev) for t in _fill(O, A, 300, 810)), norm, 0)
    assert count == 0 and loss_mean == 0
    # what comes back is a snapshot of the critic (a pickled copy in the reference), not the live module
    assert critic is not v.critic and type(critic) is type(v.critic)
    v.synchronize()
    assert torch.equal(critic.arena.data, v.critic.arena.data)
    for s in range(3):
        v.learn(indices=T(g[f"{tag}_idx"][s]), noise=T(g[f"{tag}_noise"][s]))
        v.synchronize()
        loss = v.loss_ring[s % 5].item()
        np.testing.assert_allclose(loss, g[f"{tag}_loss"][s], rtol=2e-5)
        _check_module(v.critic, g, f"{tag}_s{s}_p_")
        _check_module(v.critic_target, g, f"{tag}_s{s}_t_")
    assert v.update_count == 3 and v.opt.step.item() == int(g[f"{tag}_adam_step"])
    np.testing.assert_allclose(v.critic.state_dict()["net_q1.net.6.weight"].cpu().numpy(), g[f"{tag}_final_q1_last_w"],
                               rtol=5e-5, atol=5e-7)
    lay = v.critic.layout
    np.testing.assert_allclose(dd.summarize(lay.weight(v.opt.m, 0, 0).cpu().numpy()), g[f"{tag}_adam_m0"], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(dd.summarize(lay.weight(v.opt.v, 0, 0).cpu().numpy()), g[f"{tag}_adam_v0"], rtol=1e-4, atol=1e-12)
    np.testing.assert_allclose(v.loss_mean(), np.mean([0, 0, *g[f"{tag}_loss"]]), rtol=2e-5)   # Tracker(5) semantics


@pytest.mark.parametrize("streams", [False, True])
@pytest.mark.parametrize("distl", [False, True])
def test_p_learner_golden_trace(golden, dev, distl, streams):
    from pql_amd.algo.pql_p_learner import PQLPLearner
    from pql_amd.models.mlp import DistributionalDoubleQ, DoubleQ
    g = golden("learners"); tag = "pd" if distl else "p"; O, A = 8, 2
    p = PQLPLearner((O,), A, make_cfg(distl, streams=streams))
    p.actor.load_state_dict(_sd(dd.mlp_state(O, A, 11)))
    if distl:
        critic = DistributionalDoubleQ((O,), A, v_min=-10, v_max=10, num_atoms=51, device=dev).to(dev)
        critic.load_state_dict(_sd(dd.doubleq_state(O, A, 51, 31)))
    else:
        critic = DoubleQ((O,), A).to(dev); critic.load_state_dict(_sd(dd.doubleq_state(O, A, 1, 21)))
    norm = (T(g["learner_norm_mean"]).to(dev), T(g["learner_norm_var"]).to(dev), 1e-4)
    actor, _, count = p.update(critic, _fill(O, A, 300, 810)[0].to(dev), norm, 0)
    assert actor is not p.actor and count == 0 and (p.next_p, p.cur_capacity) == (300, 300)
    for s in range(3):
        p.learn(indices=T(g[f"{tag}_idx"][s]))
        p.synchronize()
        np.testing.assert_allclose(p.loss_ring[s % 5].item(), g[f"{tag}_loss"][s], rtol=2e-5)
        if not distl:
            _check_module(p.actor, g, f"p_s{s}_p_")
    np.testing.assert_allclose(p.actor.state_dict()["net.6.weight"].cpu().numpy(), g[f"{tag}_final_last_w"], rtol=5e-5, atol=5e-7)


def test_learn_is_noop_before_first_update(dev):
    from pql_amd.algo.pql_p_learner import PQLPLearner
    from pql_amd.algo.pql_v_learner import PQLVLearner
    v = PQLVLearner((8,), 2, make_cfg()); p = PQLPLearner((8,), 2, make_cfg())
    assert v.learn() == 0 and v.update_count == 0        # pql_v_learner.py:74
    assert p.learn() == 0.01 and p.update_count == 0     # pql_p_learner.py:48, sleep_time default 0.01


@pytest.mark.parametrize("streams", [False, True])
@pytest.mark.parametrize("hidden", [None, [512, 512, 256]])
def test_full_size_step_vs_oracle(dev, hidden, streams):
    """cfg #2 shapes (obs 88, act 16, batch 8192), reference-default and BASELINE hidden sizes: two V steps and
    two P steps with injected samples vs the CPU oracle; losses at 1e-5 relative.  Parameters: rtol 1e-5 with
    atol 1e-5 = 2 % of one Adam step (lr 5e-4): Adam divides by sqrt(v), so an entry whose gradient is ~0 turns a
    1e-9 gradient difference (fp32 summation order over 8192 rows) into a visible fraction of lr."""
    from oracle import pql_ref_cpu as ref
    from pql_amd.algo.pql_p_learner import PQLPLearner
    from pql_amd.algo.pql_v_learner import PQLVLearner
    O, A, B, cap = 88, 16, 8192, 20000
    hid = tuple(hidden) if hidden else (512, 256, 128)
    cfg = make_cfg(False, B=B, memory=cap, hidden=hidden, streams=streams)
    v = PQLVLearner((O,), A, cfg); p = PQLPLearner((O,), A, cfg)
    cst = dd.doubleq_state(O, A, 1, 21, hid); ast = dd.mlp_state(O, A, 11, hid)
    v.critic.load_state_dict(_sd(cst)); v.critic_target.arena.data.copy_(v.critic.arena.data)
    p.actor.load_state_dict(_sd(ast))
    data = _fill(O, A, cap - 100, 77)
    mean, var = T(dd.uniform((O,), 801, -0.5, 0.5)), T(dd.uniform((O,), 802, 0.5, 2.0))
    hp = ref.HyperRef(batch_size=B)
    vr = ref.VLearnerRef(O, A, hp, cap, ref.params_from_state(cst, "net_q1.net."), ref.params_from_state(cst, "net_q2.net."))
    pr = ref.PLearnerRef(O, A, hp, cap, ref.params_from_state(ast))
    vr.update(ref.params_from_state(ast), data, (mean, var, 1e-4))
    pr.update(vr.q1, vr.q2, data[0], (mean, var, 1e-4))
    norm = (mean.to(dev), var.to(dev), 1e-4)
    v.update(p.actor, tuple(t.to(dev) for t in data), norm, 0)
    p.update(v.critic, data[0].to(dev), norm, 0)
    for s in range(2):
        idx = T(dd.integers((B,), 900 + s, cap - 100)); draw = T(dd.uniform((B, A), 950 + s, -2, 2))
        lv = vr.learn(idx=idx, draw=draw)
        v.learn(indices=idx, noise=draw)
        v.synchronize()
        np.testing.assert_allclose(v.loss_ring[s % 5].item(), lv, rtol=1e-5)
        lp = pr.learn(idx=idx)
        p.learn(indices=idx)
        p.synchronize()
        np.testing.assert_allclose(p.loss_ring[s % 5].item(), lp, rtol=1e-5, atol=1e-7)
    lay = v.critic.layout
    k = 0
    for n, net in enumerate((vr.q1, vr.q2)):
        for l in range(lay.n_layers):
            np.testing.assert_allclose(lay.weight(v.critic.arena.data, n, l).cpu().numpy(), net[2 * l].detach().numpy(),
                                       rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(lay.bias(v.critic.arena.data, n, l).cpu().numpy(), net[2 * l + 1].detach().numpy(),
                                       rtol=1e-5, atol=1e-5)
    al = p.actor.layout
    for l in range(al.n_layers):
        np.testing.assert_allclose(al.weight(p.actor.arena.data, 0, l).cpu().numpy(), pr.actor[2 * l].detach().numpy(),
                                   rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(lay.weight(v.critic_target.arena.data, 0, 0).cpu().numpy(), vr.t1[0].numpy(), rtol=1e-5, atol=1e-5)


def _compare_nets(module, nets, rtol=1e-5, atol=1e-5):
    lay = module.layout
    for n, net in enumerate(nets):
        for l in range(lay.n_layers):
            np.testing.assert_allclose(lay.weight(module.arena.data, n, l).cpu().numpy(), net[2 * l].detach().numpy(),
                                       rtol=rtol, atol=atol, err_msg=f"net {n} layer {l} weight")
            np.testing.assert_allclose(lay.bias(module.arena.data, n, l).cpu().numpy(), net[2 * l + 1].detach().numpy(),
                                       rtol=rtol, atol=atol, err_msg=f"net {n} layer {l} bias")


def test_cfg4_pqld_shadowhand_shape_learner_steps_vs_oracle(dev):
    """BASELINE configs[3] at learner level: PQL-D (DistributionalDoubleQ, 51 atoms, v in [-10, 10]) at ShadowHand shape
    (obs 211, act 20), batch 8192 -- two V steps (C51 projection x2, min, BCE, backward, clip + AdamW + Polyak) and two P
    steps (DPG through the expected value of the frozen categorical critic) in the measured mode (own streams) against
    the CPU oracle on identical replay samples and noise.  Tolerances as in test_full_size_step_vs_oracle."""
    from oracle import pql_ref_cpu as ref
    from pql_amd.algo.pql_p_learner import PQLPLearner
    from pql_amd.algo.pql_v_learner import PQLVLearner
    O, A, B, cap, K = 211, 20, 8192, 20000, 51
    cfg = make_cfg(True, B=B, memory=cap, streams=True, task="ShadowHand")
    v = PQLVLearner((O,), A, cfg); p = PQLPLearner((O,), A, cfg)
    assert v.critic.num_atoms == K and v.critic.layout.dims[0] == O + A
    cst = dd.doubleq_state(O, A, K, 41); ast = dd.mlp_state(O, A, 43)
    v.critic.load_state_dict(_sd(cst)); v.critic_target.arena.data.copy_(v.critic.arena.data)
    p.actor.load_state_dict(_sd(ast))
    rows = cap - 100
    data = list(_fill(O, A, rows, 177))
    data[2] = T(dd.uniform((rows, 1), 179, -2.0, 2.0))    # rewards wide enough to move mass across several atoms
    data = tuple(data)
    mean, var = T(dd.uniform((O,), 811, -0.5, 0.5)), T(dd.uniform((O,), 812, 0.5, 2.0))
    hp = ref.HyperRef(batch_size=B, distl=True)
    vr = ref.VLearnerRef(O, A, hp, cap, ref.params_from_state(cst, "net_q1.net."), ref.params_from_state(cst, "net_q2.net."))
    pr = ref.PLearnerRef(O, A, hp, cap, ref.params_from_state(ast))
    vr.update(ref.params_from_state(ast), data, (mean, var, 1e-4))
    pr.update(vr.q1, vr.q2, data[0], (mean, var, 1e-4))
    norm = (mean.to(dev), var.to(dev), 1e-4)
    critic, _, _ = v.update(p.actor, tuple(t.to(dev) for t in data), norm, 0)
    p.update(critic, data[0].to(dev), norm, 0)
    for s in range(2):
        idx = T(dd.integers((B,), 920 + s, rows)); draw = T(dd.uniform((B, A), 970 + s, -2, 2))
        lv = vr.learn(idx=idx, draw=draw)
        v.learn(indices=idx, noise=draw)
        v.synchronize()
        np.testing.assert_allclose(v.loss_ring[s % 5].item(), lv, rtol=2e-5)
        lp = pr.learn(idx=idx)
        p.learn(indices=idx)
        p.synchronize()
        np.testing.assert_allclose(p.loss_ring[s % 5].item(), lp, rtol=2e-5, atol=1e-6)
    _compare_nets(v.critic, (vr.q1, vr.q2))
    _compare_nets(v.critic_target, (vr.t1, vr.t2))
    _compare_nets(p.actor, (pr.actor,))


def test_cfg5_humanoid_nstep5_batch32768_v_step_vs_oracle(dev):
    """BASELINE configs[4] at learner level: obs 108 / act 21, n-step 5 (gamma^5 in the TD target), batch 32768 -- one V step
    and one P step against the CPU oracle (own streams)."""
    from oracle import pql_ref_cpu as ref
    from pql_amd.algo.pql_p_learner import PQLPLearner
    from pql_amd.algo.pql_v_learner import PQLVLearner
    O, A, B, cap = 108, 21, 32768, 50000
    cfg = make_cfg(False, B=B, memory=cap, nstep=5, streams=True, task="Humanoid")
    v = PQLVLearner((O,), A, cfg); p = PQLPLearner((O,), A, cfg)
    cst = dd.doubleq_state(O, A, 1, 51); ast = dd.mlp_state(O, A, 53)
    v.critic.load_state_dict(_sd(cst)); v.critic_target.arena.data.copy_(v.critic.arena.data)
    p.actor.load_state_dict(_sd(ast))
    rows = cap - 10
    data = _fill(O, A, rows, 277)
    mean, var = T(dd.uniform((O,), 821, -0.5, 0.5)), T(dd.uniform((O,), 822, 0.5, 2.0))
    hp = ref.HyperRef(batch_size=B, nstep=5)
    vr = ref.VLearnerRef(O, A, hp, cap, ref.params_from_state(cst, "net_q1.net."), ref.params_from_state(cst, "net_q2.net."))
    pr = ref.PLearnerRef(O, A, hp, cap, ref.params_from_state(ast))
    vr.update(ref.params_from_state(ast), data, (mean, var, 1e-4))
    pr.update(vr.q1, vr.q2, data[0], (mean, var, 1e-4))
    norm = (mean.to(dev), var.to(dev), 1e-4)
    critic, _, _ = v.update(p.actor, tuple(t.to(dev) for t in data), norm, 0)
    p.update(critic, data[0].to(dev), norm, 0)
    idx = T(dd.integers((B,), 930, rows)); draw = T(dd.uniform((B, A), 980, -2, 2))
    lv = vr.learn(idx=idx, draw=draw)
    v.learn(indices=idx, noise=draw)
    v.synchronize()
    np.testing.assert_allclose(v.loss_ring[0].item(), lv, rtol=1e-5)
    lp = pr.learn(idx=idx)
    p.learn(indices=idx)
    p.synchronize()
    np.testing.assert_allclose(p.loss_ring[0].item(), lp, rtol=1e-5, atol=1e-7)
    _compare_nets(v.critic, (vr.q1, vr.q2))
    _compare_nets(v.critic_target, (vr.t1, vr.t2))
    _compare_nets(p.actor, (pr.actor,))


@pytest.mark.parametrize("distl", [False, True])
def test_fused_tail_matches_the_separate_launches(dev, distl):
    """algo.fused_tail folds the loss reduction into the optimiser launch and clip_grad_norm_'s sum of squares into backward's
    slab reduction (2 launches fewer per step).  The loss fold is the same sum in the same order (bit-equal ring); the norm's
    partial sums are grouped differently (last-bit differences in the clip factor), so parameters agree to ~1e-6 relative,
    as with the data-parallel path, which still uses the stand-alone launches."""
    from pql_amd.algo.pql_p_learner import PQLPLearner
    from pql_amd.algo.pql_v_learner import PQLVLearner
    O, A, B, cap, K = 88, 16, 4096, 8000, 51
    outs = []
    for tail in (True, False):
        cfg = make_cfg(distl, B=B, memory=cap, hidden=[512, 512, 256])
        cfg.algo.fused_tail = tail
        cfg.algo.td_in_head = False   # (the TD-in-head form groups the loss partials differently: its own test below)
        v = PQLVLearner((O,), A, cfg); p = PQLPLearner((O,), A, cfg)
        assert v._fused_tail is tail and p._fused_tail is tail
        v.critic.load_state_dict(_sd(dd.doubleq_state(O, A, K if distl else 1, 61, (512, 512, 256))))
        v.critic_target.arena.data.copy_(v.critic.arena.data)
        p.actor.load_state_dict(_sd(dd.mlp_state(O, A, 63, (512, 512, 256))))
        data = tuple(t.to(dev) for t in _fill(O, A, cap - 7, 377))
        norm = (T(dd.uniform((O,), 831, -0.5, 0.5)).to(dev), T(dd.uniform((O,), 832, 0.5, 2.0)).to(dev), 1e-4)
        critic, _, _ = v.update(p.actor, data, norm, 0)
        p.update(critic, data[0], norm, 0)
        for s in range(3):
            idx = T(dd.integers((B,), 940 + s, cap - 7)); draw = T(dd.uniform((B, A), 990 + s, -2, 2))
            v.learn(indices=idx, noise=draw)
            p.learn(indices=idx)
        torch.cuda.synchronize()
        outs.append((v.critic.arena.data.clone(), v.critic_target.arena.data.clone(), v.opt.m.clone(), v.opt.v.clone(),
                     v.loss_ring.clone(), v.opt.gnorm.clone(), p.actor.arena.data.clone(), p.opt.v.clone(), p.loss_ring.clone(),
                     v.opt.step.clone(), p.opt.step.clone()))
    assert int(outs[0][-1]) == 3 and int(outs[0][-2]) == 3
    for k, (a, b) in enumerate(zip(*outs)):
        if k in (4, 8):   # loss rings: only the first step's loss is computed from bit-equal parameters
            s0 = int(outs[0][9]) - 3   # ring slot of the first step
            assert torch.equal(a[s0 % 5], b[s0 % 5])
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-7)
        else:
            torch.testing.assert_close(a.float(), b.float(), rtol=2e-5, atol=2e-6)
    assert float(outs[0][4].abs().sum()) > 0 and float(outs[0][8].abs().sum()) > 0   # the folded losses did land in the rings


def test_td_in_head_matches_the_separate_loss_launch(dev):
    """algo.td_in_head: the scalar twin critic's TD target, MSE loss and dL/dQ are formed inside the head's backward pass
    (k_skinny_bwd<1, CH, true>) instead of by pqlk_td_mse_loss + a dL/dQ round trip through memory.  Per sample the same
    arithmetic in the same order, so the gradient -- and with it every parameter -- is BIT-equal; the loss value sums
    its partials in different groups (64-row blocks per net instead of 256-row blocks) and agrees to 1e-6 relative."""
    from pql_amd.algo.pql_v_learner import PQLVLearner
    from pql_amd.models.mlp import TanhMLPPolicy
    O, A, B, cap = 88, 16, 4096, 8000
    actor = TanhMLPPolicy((O,), A, hidden_layers=[512, 512, 256]).to(dev)
    actor.load_state_dict(_sd(dd.mlp_state(O, A, 63, (512, 512, 256))))
    outs = []
    for td in (True, False):
        cfg = make_cfg(False, B=B, memory=cap, hidden=[512, 512, 256])
        cfg.algo.td_in_head = td
        cfg.algo.td_in_forward = False   # (the head's backward inside the FORWARD launch sums the head's partials in other groups: next test)
        v = PQLVLearner((O,), A, cfg)
        v.critic.load_state_dict(_sd(dd.doubleq_state(O, A, 1, 61, (512, 512, 256))))
        v.critic_target.arena.data.copy_(v.critic.arena.data)
        data = tuple(t.to(dev) for t in _fill(O, A, cap - 7, 377))
        norm = (T(dd.uniform((O,), 831, -0.5, 0.5)).to(dev), T(dd.uniform((O,), 832, 0.5, 2.0)).to(dev), 1e-4)
        v.update(actor, data, norm, 0)
        for s in range(3):
            v.learn(indices=T(dd.integers((B,), 940 + s, cap - 7)), noise=T(dd.uniform((B, A), 990 + s, -2, 2)))
        torch.cuda.synchronize()
        assert (v._ws["td_parts"] > 0) is td
        outs.append((v.critic.arena.data.clone(), v.critic_target.arena.data.clone(), v.opt.m.clone(), v.opt.v.clone(), v.opt.gnorm.clone(),
                     v._ws["grads"].clone(), v.loss_ring.clone()))
    for a, b in zip(outs[0][:-1], outs[1][:-1]):
        assert torch.equal(a, b)
    torch.testing.assert_close(outs[0][-1], outs[1][-1], rtol=1e-6, atol=1e-9)
    assert float(outs[0][-1].abs().sum()) > 0


@pytest.mark.parametrize("hidden,B", [([512, 512, 256], 4096), ([512, 256, 128], 1000), ([512, 128], 96), ([1024, 512], 300)])
def test_td_in_forward_matches_the_head_backward_launch(dev, hidden, B):
    """algo.td_in_forward: the critic's fused forward forms the TD error from the Q it has just computed and leaves the head's
    whole backward (dL/dZ of the last hidden layer, the head's dW / db partials, the loss partials) while that layer's
    activations are still in LDS, instead of k_skinny_bwd<1, CH, true> re-reading them in a launch of its own.  Per element
    the same arithmetic: dL/dZ and with it the gradient of every hidden layer are BIT-equal after the first step; the head's
    own dW / db and the loss add their per-block partials in other groups (64- or 32-row tiles instead of 32-row blocks) and
    agree to 1e-5 relative; three steps on, parameters agree to that order.  Full, ragged and tiny batches; 64- and 32-row tiles."""
    from pql_amd.algo.pql_v_learner import PQLVLearner
    from pql_amd.models.mlp import TanhMLPPolicy
    O, A, cap = 88, 16, 8000
    actor = TanhMLPPolicy((O,), A, hidden_layers=hidden).to(dev)
    actor.load_state_dict(_sd(dd.mlp_state(O, A, 63, tuple(hidden))))
    outs = []
    for fwd in (True, False):
        cfg = make_cfg(False, B=B, memory=cap, hidden=hidden)
        cfg.algo.td_in_forward = fwd
        v = PQLVLearner((O,), A, cfg)
        v.critic.load_state_dict(_sd(dd.doubleq_state(O, A, 1, 61, tuple(hidden))))
        v.critic_target.arena.data.copy_(v.critic.arena.data)
        data = tuple(t.to(dev) for t in _fill(O, A, cap - 7, 377))
        norm = (T(dd.uniform((O,), 831, -0.5, 0.5)).to(dev), T(dd.uniform((O,), 832, 0.5, 2.0)).to(dev), 1e-4)
        v.update(actor, data, norm, 0)
        v.learn(indices=T(dd.integers((B,), 940, cap - 7)), noise=T(dd.uniform((B, A), 990, -2, 2)))
        torch.cuda.synchronize()
        assert (v._ws["td_fwd"] > 0) is fwd and v._ws["td_parts"] > 0
        g1, lay = v._ws["grads"].clone(), v.critic.layout
        for s in range(1, 3):
            v.learn(indices=T(dd.integers((B,), 940 + s, cap - 7)), noise=T(dd.uniform((B, A), 990 + s, -2, 2)))
        torch.cuda.synchronize()
        outs.append((g1, v.critic.arena.data.clone(), v.loss_ring.clone(), lay))
    (ga, pa, la, lay), (gb, pb, lb, _) = outs
    nl = lay.n_layers
    for n in range(2):
        for l in range(nl - 1):
            assert torch.equal(lay.weight(ga, n, l), lay.weight(gb, n, l)) and torch.equal(lay.bias(ga, n, l), lay.bias(gb, n, l)), (n, l)
        wa, wb = lay.weight(ga, n, nl - 1), lay.weight(gb, n, nl - 1)
        torch.testing.assert_close(wa, wb, rtol=1e-5, atol=1e-6 * float(wb.abs().max()))
        torch.testing.assert_close(lay.bias(ga, n, nl - 1), lay.bias(gb, n, nl - 1), rtol=1e-5, atol=1e-8)
        assert float(wb.abs().max()) > 0
    torch.testing.assert_close(la, lb, rtol=1e-6, atol=1e-9)
    torch.testing.assert_close(pa, pb, rtol=1e-4, atol=2e-6)
    assert float(la.abs().sum()) > 0


def test_learn_many_replays_a_whole_run_as_one_graph_with_the_bits_of_the_per_step_calls(dev):
    """`learn_many(n)` = n `learn()` calls.  With n = the draws-ahead depth at the start of a run (what the fixed-ratio loop issues
    between two `update()` calls) the steps replay as ONE hipGraph per learner; a run already begun, another n, or per-step draws
    fall back to the loop of `learn()`.  Same seeds -> bit-identical parameters, Adam state, loss rings and generator offsets
    against the per-step calls, across updates, for both learners."""
    from pql_amd.algo.pql_p_learner import PQLPLearner
    from pql_amd.algo.pql_v_learner import PQLVLearner
    from pql_amd.models.mlp import TanhMLPPolicy
    O, A, B = 24, 6, 512
    res = {}
    for many in (False, True):
        cfg = make_cfg(False, B=B, memory=4000, hidden=[256, 128], graph=True)
        cfg.algo.rng = "auto"
        v, p = PQLVLearner((O,), A, cfg), PQLPLearner((O,), A, cfg)
        v.critic.load_state_dict(_sd(dd.doubleq_state(O, A, 1, 21, hidden=(256, 128)))); v.critic_target.arena.data.copy_(v.critic.arena.data)
        p.actor.load_state_dict(_sd(dd.mlp_state(O, A, 11, hidden=(256, 128))))
        v.use_private_rng(5); p.use_private_rng(6)
        from pql_amd.utils import rng as R
        if R.verified(dev) is None:
            pytest.skip("pqlk_philox_draws does not reproduce torch's draws on this device: no draws-ahead runs")
        Kv, Kp = v._depth, p._depth
        runs_v = runs_p = 0
        for it in range(4):
            norm = (T(dd.uniform((O,), 40 + it, -0.5, 0.5)).to(dev), T(dd.uniform((O,), 50 + it, 0.5, 2.0)).to(dev), 1e-4)
            data = tuple(t.to(dev) for t in _fill(O, A, 700, 100 + it))
            critic, _, _ = v.update(p.actor, data, norm, 0)
            p.update(critic, data[0], norm, 0)
            # iteration 2 issues a PARTIAL run first (3 steps), then the rest: learn_many must fall back to per-step calls there
            plan_v = [Kv] if it != 2 else [3, Kv - 3]
            for n in plan_v:
                if many:
                    runs_v += v._run_in_one_graph(v._workspace(B), n)
                    v.learn_many(n)
                else:
                    for _ in range(n):
                        v.learn()
            for n in ([Kp] if it != 2 else [1, Kp - 1]):   # (a partial run prefetches only its own steps' draws and rows)
                if many:
                    p.learn_many(n)
                else:
                    for _ in range(n):
                        p.learn()
        torch.cuda.synchronize()
        assert v.update_count == 4 * Kv and p.update_count == 4 * Kp
        if many:
            assert runs_v == 3 and v._run_graph is not None and p._run_graph is not None   # three whole runs, one split run
        res[many] = [t.clone() for t in (v.critic.arena.data, v.critic_target.arena.data, v.opt.m, v.opt.v, v.loss_ring, p.actor.arena.data,
                                         p.opt.m, p.opt.v, p.loss_ring)] + [torch.tensor([v.gen.get_offset(), p.gen.get_offset()])]
    for a, b in zip(res[False], res[True]):
        assert torch.equal(a, b)
    assert float(res[True][4].abs().sum()) > 0 and float(res[True][8].abs().sum()) > 0


@pytest.mark.parametrize("graph", [False, True])
def test_draws_ahead_and_batched_gather_equal_the_per_step_torch_draws(dev, graph):
    """algo.rng=auto: the draws of the next K steps come from one pqlk_philox_draws launch and their K x B rows from one gather,
    instead of randint + normal_ + gather per step (algo.rng=torch, the reference's call pattern).  Same seeds
    -> the SAME indices and noise, hence bit-identical parameters, optimiser state and loss rings for both learners -- across
    `update()` calls that arrive in the middle of a prefetched run (3 steps, update, 8 steps, update, 5 steps: the leftovers are
    dropped and re-drawn at the generator's offset) and with the step replayed from per-slot hipGraphs."""
    from pql_amd.algo.pql_p_learner import PQLPLearner
    from pql_amd.algo.pql_v_learner import PQLVLearner
    O, A, B, cap = 8, 2, 256, 3000
    outs = []
    for mode in ("torch", "auto"):
        cfg = make_cfg(False, B=B, memory=cap, graph=graph)
        cfg.algo.rng = mode
        v, p = PQLVLearner((O,), A, cfg), PQLPLearner((O,), A, cfg)
        assert v._depth == 8 and p._depth == 4
        v.critic.load_state_dict(_sd(dd.doubleq_state(O, A, 1, 21))); v.critic_target.arena.data.copy_(v.critic.arena.data)
        p.actor.load_state_dict(_sd(dd.mlp_state(O, A, 11)))
        v.use_private_rng(1234); p.use_private_rng(4321)
        norm = (T(dd.uniform((O,), 6, -0.5, 0.5)).to(dev), T(dd.uniform((O,), 7, 0.5, 2.0)).to(dev), 1e-4)
        idxs = []
        for phase, steps in enumerate((3, 8, 5)):
            data = tuple(t.to(dev) for t in _fill(O, A, 700, 50 + 10 * phase))      # the ring grows: the randint bound changes
            critic, _, _ = v.update(p.actor, data, norm, 0)
            p.update(critic, data[0], norm, 0)
            for k in range(steps):
                v.learn()
                if k % 2 == 1:
                    p.learn()
                idxs.append(v._ahead.idx[(v._ahead.pos - 1)].clone() if v._ahead is not None else v._ws["idx"].clone())
        torch.cuda.synchronize()
        assert v.rng == ("philox" if mode == "auto" else "torch") == p.rng
        outs.append((v.critic.arena.data.clone(), v.critic_target.arena.data.clone(), v.opt.m.clone(), v.opt.v.clone(), v.loss_ring.clone(),
                     p.actor.arena.data.clone(), p.opt.m.clone(), p.loss_ring.clone(), torch.stack(idxs), v.gen.get_offset(), p.gen.get_offset()))
    for a, b in zip(*outs):
        assert torch.equal(a, b) if torch.is_tensor(a) else a == b
    assert v.update_count == 16 and p.update_count == 7   # (1 + 4 + 2 P-steps)


@pytest.mark.parametrize("graph", [False, True])
def test_draws_ahead_tiles_follow_data_changed_behind_the_learners_back(dev, graph):
    """The tiles gathered ahead (algo.rng=auto) must not outlive the data they were gathered from when that data changes WITHOUT an
    `update()` call: rows inserted straight through `memory.add_to_buffer` (the ring's insert counter and the randint bound move)
    and `normalize_tuple` assigned from outside -- both legal on the reference's objects (pql_v_learner.py:117-122 does nothing
    else).  With the same seeds the learner must then do exactly what the per-step learner (algo.rng=torch, no tiles ahead) does:
    bit-identical parameters, loss rings, sample indices and generator offsets."""
    from pql_amd.algo.pql_v_learner import PQLVLearner
    from pql_amd.models.mlp import TanhMLPPolicy
    O, A, B, cap = 8, 2, 256, 3000
    actor = TanhMLPPolicy((O,), A).to(dev); actor.load_state_dict(_sd(dd.mlp_state(O, A, 11)))
    outs = []
    for mode in ("torch", "auto"):
        cfg = make_cfg(False, B=B, memory=cap, graph=graph)
        cfg.algo.rng = mode
        v = PQLVLearner((O,), A, cfg)
        v.critic.load_state_dict(_sd(dd.doubleq_state(O, A, 1, 21))); v.critic_target.arena.data.copy_(v.critic.arena.data)
        v.use_private_rng(1234)
        norm = (T(dd.uniform((O,), 6, -0.5, 0.5)).to(dev), T(dd.uniform((O,), 7, 0.5, 2.0)).to(dev), 1e-4)
        norm2 = (T(dd.uniform((O,), 16, -0.5, 0.5)).to(dev), T(dd.uniform((O,), 17, 0.5, 2.0)).to(dev), 1e-4)
        v.update(actor, tuple(t.to(dev) for t in _fill(O, A, 700, 50)), norm, 0)
        idxs = []

        def steps(n):
            for _ in range(n):
                v.learn()
                idxs.append(v._ahead.idx[(v._ahead.pos - 1)].clone() if v._ahead is not None else v._ws["idx"].clone())
        steps(3)                                                               # 5 more steps are prepared ahead (auto)
        v.memory.add_to_buffer(tuple(t.to(dev) for t in _fill(O, A, 500, 60)))   # new rows + a larger randint bound
        steps(2)
        v.normalize_tuple = norm2                                              # new statistics
        steps(4)
        v.learn_many(8)                                                        # a whole run, nothing stale left
        torch.cuda.synchronize()
        assert v.rng == ("philox" if mode == "auto" else "torch")
        outs.append((v.critic.arena.data.clone(), v.critic_target.arena.data.clone(), v.opt.m.clone(), v.loss_ring.clone(), torch.stack(idxs),
                     v.gen.get_offset()))
    for a, b in zip(*outs):
        assert torch.equal(a, b) if torch.is_tensor(a) else a == b
    assert int(outs[0][4][3:].max()) >= 700      # steps after the direct insert really sampled the new rows


@pytest.mark.parametrize("graph", [False, True])
def test_p_learner_launch_sequences_agree(dev, graph):
    """The three forms of the P-learner step -- round 3's 16 launches (`algo.dpg_fused=False`), round 4's 13 with the critic reading
    a concatenated [obs | action] tile (`algo.dpg_split_input=False`) and with the critic reading the actor's input tile + output
    block (default) -- on the same seeds and data, AllegroHand shapes at the BASELINE hidden sizes.  The two 13-launch forms run
    the same kernels on the same values (only where the critic's input rows come from differs): BIT-identical parameters, Adam
    state and loss ring.  The 16-launch form sums the actor head's dW / db and the loss in another order: equal to rounding."""
    from pql_amd.algo.pql_p_learner import PQLPLearner
    from pql_amd.models.mlp import DoubleQ
    O, A, B, cap, hidden = 88, 16, 2048, 6000, [512, 512, 256]
    critic = DoubleQ((O,), A, hidden_layers=hidden).to(dev)
    critic.load_state_dict(_sd(dd.doubleq_state(O, A, 1, 21, hidden=hidden)))
    norm = (T(dd.uniform((O,), 6, -0.5, 0.5)).to(dev), T(dd.uniform((O,), 7, 0.5, 2.0)).to(dev), 1e-4)
    outs = {}
    for name, ov in (("split", {}), ("cat", {"dpg_split_input": False}), ("r3", {"dpg_fused": False})):
        cfg = make_cfg(False, B=B, memory=cap, hidden=hidden, graph=graph, task="AllegroHand")
        for k, v_ in ov.items():
            cfg.algo[k] = v_
        p = PQLPLearner((O,), A, cfg)
        p.actor.load_state_dict(_sd(dd.mlp_state(O, A, 11, hidden=hidden)))
        p.use_private_rng(4321)
        p.update(critic, T(dd.uniform((5000, O), 33, -3, 3)).to(dev), norm, 0)
        for _ in range(2):
            p.learn_many(4)
        p.learn()
        torch.cuda.synchronize()
        ws = p._ws
        assert ws["dpg_fused"] == (name != "r3") and ws["split_in"] == (name == "split")
        outs[name] = (p.actor.arena.data.clone(), p.opt.m.clone(), p.opt.v.clone(), p.loss_ring.clone(), p.gen.get_offset())
    for a, b in zip(outs["split"], outs["cat"]):
        assert torch.equal(a, b) if torch.is_tensor(a) else a == b
    ref_, new = outs["r3"], outs["split"]
    assert ref_[4] == new[4]
    torch.testing.assert_close(new[3], ref_[3], rtol=1e-5, atol=1e-7)                 # losses of the last five steps
    # nine AdamW steps of lr 5e-4 on gradients that differ in the last bits: the arenas agree in norm to 1e-4 and on average to 1e-6
    # (single elements whose gradient is rounding noise may move by a fraction of lr per step: see _check_module)
    assert float((new[0] - ref_[0]).norm() / ref_[0].norm()) < 1e-4
    assert float((new[0] - ref_[0]).abs().mean()) < 1e-6


def test_graph_replay_matches_eager(dev):
    """hipGraph-captured learn() is the same launch sequence: with equal seeds it must reproduce the eager
    parameters bit for bit (same kernels, same order, same RNG offsets)."""
    from pql_amd.algo.pql_v_learner import PQLVLearner
    from pql_amd.models.mlp import TanhMLPPolicy
    O, A, B, cap = 8, 2, 256, 2000
    actor = TanhMLPPolicy((O,), A).to(dev); actor.load_state_dict(_sd(dd.mlp_state(O, A, 11)))
    outs = []
    for graph in (False, True):
        v = PQLVLearner((O,), A, make_cfg(False, B=B, memory=cap, graph=graph))
        v.critic.load_state_dict(_sd(dd.doubleq_state(O, A, 1, 21))); v.critic_target.arena.data.copy_(v.critic.arena.data)
        v.update(actor, tuple(t.to(dev) for t in _fill(O, A, cap, 5)), None, 0)
        v.use_private_rng(1234)
        for _ in range(4):
            v.learn()
        torch.cuda.synchronize()
        outs.append((v.critic.arena.data.clone(), v.critic_target.arena.data.clone(), v.loss_ring.clone(), v.opt.step.item()))
    assert outs[0][3] == outs[1][3] == 4
    for a, b in zip(outs[0][:3], outs[1][:3]):
        assert torch.equal(a, b)


# --------------------------------------------------------------------------- DDPG (BASELINE cfg #1) + entry points
def _ddpg_cfg(extra=()):
    from pql_amd.utils.cfg import load_cfg
    return load_cfg(["algo=ddpg_algo", "task.name=Toy", "num_envs=64", "algo.batch_size=256", "algo.memory_size=100000",
                     "device=cuda:0", "sim_device=cuda:0", *extra])


def test_ddpg_update_vs_oracle(dev):
    """cfg #1 shapes (64 envs, obs 8, act 2, batch 256, replay 100k): three inner iterations of update_net with
    injected samples/noise against the oracle's DDPGRef (un-clamped normalisation, critic step, actor step through
    the updated critic, Polyak)."""
    from oracle import pql_ref_cpu as ref
    from pql_amd.algo.ddpg import AgentDDPG
    from pql_amd.envs.synthetic import create_task_env
    from pql_amd.replay.simple_replay import ReplayBuffer
    cfg = _ddpg_cfg()
    agent = AgentDDPG(create_task_env(cfg), cfg)
    O, A, B, rows = 8, 2, 256, 3000
    cst, ast = dd.doubleq_state(O, A, 1, 21), dd.mlp_state(O, A, 11)
    agent.critic.load_state_dict(_sd(cst)); agent.critic_target.arena.data.copy_(agent.critic.arena.data)
    agent.actor.load_state_dict(_sd(ast))
    mean, var = T(dd.uniform((O,), 801, -0.5, 0.5)), T(dd.uniform((O,), 802, 0.5, 2.0))
    agent.obs_rms.mean, agent.obs_rms.var = mean.to(dev), var.to(dev)
    data = _fill(O, A, rows, 77)
    memory = ReplayBuffer(100000, (O,), A, device=dev)
    memory.add_to_buffer(tuple(t.to(dev) for t in data))
    orc = ref.DDPGRef(O, A, ref.HyperRef(batch_size=B), 100000, ref.params_from_state(ast),
                      ref.params_from_state(cst, "net_q1.net."), ref.params_from_state(cst, "net_q2.net."))
    orc.ring.insert(*data); orc.norm = (mean, var, 1e-4)
    for s in range(3):
        idx, draw = T(dd.integers((B,), 40 + s, rows)), T(dd.uniform((B, A), 50 + s, -2, 2))
        cl, al_ = orc.update_once(idx, draw)
        agent.update_once(memory, indices=idx, noise=draw)
        np.testing.assert_allclose(agent.closs[s % 5].item(), cl, rtol=2e-5)
        np.testing.assert_allclose(agent.aloss[s % 5].item(), al_, rtol=2e-5, atol=1e-7)
    lay = agent.critic.layout
    for n, net in enumerate((orc.q1, orc.q2)):
        for l in range(lay.n_layers):
            np.testing.assert_allclose(lay.weight(agent.critic.arena.data, n, l).cpu().numpy(), net[2 * l].detach().numpy(),
                                       rtol=1e-5, atol=1e-5)
    for l in range(agent.actor.layout.n_layers):
        np.testing.assert_allclose(agent.actor.layout.weight(agent.actor.arena.data, 0, l).cpu().numpy(),
                                   orc.actor[2 * l].detach().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(lay.weight(agent.critic_target.arena.data, 1, 1).cpu().numpy(), orc.t2[2].numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("tag", ["tgt0", "tgt1"])
def test_ddpg_golden_trace(golden, dev, tag):
    """AgentDDPG.update_once (HIP launch sequence) vs three iterations of the reference's own AgentDDPG.update_critic / update_actor /
    soft_update (tests/golden/ddpg.npz, pql/algo/ddpg.py:119-166) on identical samples and target-policy noise: losses, every
    parameter tensor of the actor, the critic and the targets.  tgt0: no_tgt_actor=True (every shipped config); tgt1:
    no_tgt_actor=False -- a Polyak-averaged target actor that starts away from the actor (ddpg.py:21-22,134-135)."""
    from pql_amd.algo.ddpg import AgentDDPG
    from pql_amd.envs.synthetic import create_task_env
    from pql_amd.replay.simple_replay import ReplayBuffer
    g = golden("ddpg"); O, A = 8, 2
    cfg = _ddpg_cfg(["algo.batch_size=64", "algo.memory_size=400", f"algo.no_tgt_actor={tag == 'tgt0'}"])
    agent = AgentDDPG(create_task_env(cfg), cfg)
    agent.actor.load_state_dict(_sd(dd.mlp_state(O, A, 11)))
    agent.critic.load_state_dict(_sd(dd.doubleq_state(O, A, 1, 21))); agent.critic_target.arena.data.copy_(agent.critic.arena.data)
    assert (agent.actor_target is agent.actor) == (tag == "tgt0")
    if tag == "tgt1":
        agent.actor_target.load_state_dict(_sd(dd.mlp_state(O, A, 13)))
    agent.obs_rms.mean, agent.obs_rms.var = T(g["ddpg_norm_mean"]).to(dev), T(g["ddpg_norm_var"]).to(dev)
    memory = ReplayBuffer(400, (O,), A, device=dev)
    memory.add_to_buffer(tuple(t.to(dev) for t in _fill(O, A, 300, 810)))
    for s in range(3):
        agent.update_once(memory, indices=T(g[f"ddpg_{tag}_idx"][s]), noise=T(g[f"ddpg_{tag}_noise"][s]))
        np.testing.assert_allclose(agent.closs[s % 5].item(), g[f"ddpg_{tag}_closs"][s], rtol=2e-5)
        np.testing.assert_allclose(agent.aloss[s % 5].item(), g[f"ddpg_{tag}_aloss"][s], rtol=2e-5)
        _check_module(agent.actor, g, f"ddpg_{tag}_s{s}_a_")
        _check_module(agent.critic, g, f"ddpg_{tag}_s{s}_c_")
        _check_module(agent.critic_target, g, f"ddpg_{tag}_s{s}_t_")
        if tag == "tgt1":
            _check_module(agent.actor_target, g, f"ddpg_{tag}_s{s}_at_")
    np.testing.assert_allclose(agent.actor.layout.weight(agent.actor.arena.data, 0, 3).cpu().numpy(), g[f"ddpg_{tag}_final_actor_last_w"],
                               rtol=5e-5, atol=5e-7)
    np.testing.assert_allclose(agent.critic_target.layout.weight(agent.critic_target.arena.data, 0, 3).cpu().numpy(),
                               g[f"ddpg_{tag}_final_tq1_last_w"], rtol=5e-5, atol=5e-7)
    if tag == "tgt1":
        np.testing.assert_allclose(agent.actor_target.layout.weight(agent.actor_target.arena.data, 0, 3).cpu().numpy(),
                                   g[f"ddpg_{tag}_final_tactor_last_w"], rtol=5e-5, atol=5e-7)


@pytest.mark.parametrize("algo", ["sac", "crossq"])
def test_target_actor_is_polyak_averaged_in_sac_and_crossq(dev, algo):
    """no_tgt_actor=False (sac.py:19,104-105; crossQ.py:21,132-133): the target policy is a separate arena that follows the policy by
    theta' <- tau theta + (1 - tau) theta' after every update; CrossQ takes its target-policy actions from it."""
    from pql_amd.envs.synthetic import create_task_env
    from pql_amd.replay.simple_replay import ReplayBuffer
    if algo == "sac":
        from pql_amd.algo.sac import AgentSAC as Agent
        cfg = _sac_cfg(["algo.no_tgt_actor=False"])
    else:
        from pql_amd.algo.crossq import AgentCrossQ as Agent
        cfg = _crossq_cfg(["algo.no_tgt_actor=False"])
    agent = Agent(create_task_env(cfg), cfg)
    assert agent.actor_target is not agent.actor and torch.equal(agent.actor_target.arena.data, agent.actor.arena.data)
    O, A = 8, 2
    memory = ReplayBuffer(400, (O,), A, device=dev)
    memory.add_to_buffer(tuple(t.to(dev) for t in _fill(O, A, 300, 810)))
    tau = float(cfg.algo.tau)
    for _ in range(2):
        before = agent.actor_target.arena.data.clone()
        agent.update_once(memory)
        want = agent.actor.arena.data * tau + before * (1.0 - tau)     # soft_update, torch_util.py:9-12
        torch.testing.assert_close(agent.actor_target.arena.data, want, rtol=0, atol=0)
    assert not torch.equal(agent.actor_target.arena.data, agent.actor.arena.data)


def test_train_baselines_entry_point_cfg1(dev):
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("train_baselines", os.path.join(root, "scripts", "train_baselines.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    out = mod.main(_ddpg_cfg(["max_step=4000", "algo.update_times=8"]))
    assert out["global_steps"] > 4000 and np.isfinite(out["train/critic_loss"]) and np.isfinite(out["train/actor_loss"])


@pytest.mark.parametrize("distl", [False, True])
def test_train_pql_entry_point(dev, distl):
    """scripts/train_pql.py on a tiny config: the loop keeps the reference's counters and design ratios
    (8 critic steps and 4 actor steps per env iteration) and finite losses."""
    import importlib.util, os
    from pql_amd.utils.cfg import load_cfg
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("train_pql", os.path.join(root, "scripts", "train_pql.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    cfg = load_cfg(["task.name=Toy", "num_envs=64", "algo.batch_size=256", "algo.memory_size=20000", "algo.num_gpus=1",
                    f"algo.distl={distl}", "max_step=6000", "algo.graph=True"])
    out = mod.main(cfg)
    iters = (out["global_steps"] - 64 * 32) // 64
    assert out["critic_updates"] == 8 * iters and out["actor_updates"] == 4 * iters


# --------------------------------------------------------------------------- data parallel on the GPU (2 ranks, one card)
def _dp_rank(rank, world, port, ret):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)   # rehearsal backend; production uses RCCL ("nccl")
    try:
        from pql_amd.algo.pql_v_learner import PQLVLearner
        from pql_amd.models.mlp import TanhMLPPolicy
        dev = torch.device("cuda:0")
        O, A, B = 8, 2, 64
        g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "learners.npz")))
        outs = {}
        for graph in (False, True):
            v = PQLVLearner((O,), A, make_cfg(False, B=B // world, graph=graph), process_group=dist.group.WORLD)
            v.critic.load_state_dict(_sd(dd.doubleq_state(O, A, 1, 21))); v.critic_target.arena.data.copy_(v.critic.arena.data)
            actor = TanhMLPPolicy((O,), A).to(dev); actor.load_state_dict(_sd(dd.mlp_state(O, A, 11)))
            norm = (T(g["learner_norm_mean"]).to(dev), T(g["learner_norm_var"]).to(dev), 1e-4)
            v.update(actor, tuple(t.to(dev) for t in _fill(O, A, 300, 810)), norm, 0)
            if not graph:   # rank r takes idx[r*B/G:(r+1)*B/G] of the golden single-GPU batch (SURVEY 8e parity rule)
                sl = slice(rank * B // world, (rank + 1) * B // world)
                v.learn(indices=T(g["v_idx"][0][sl]), noise=T(g["v_noise"][0][sl]))
                outs["injected"] = dd.summarize(v.critic.layout.weight(v.critic.arena.data, 0, 1).cpu().numpy())
            else:           # graph mode: two captured graphs around the eager all-reduce; replicas must stay identical
                v.use_private_rng(99)
                for _ in range(3):
                    v.learn()
                torch.cuda.synchronize()
                outs["graph"] = v.critic.arena.data.cpu().numpy()
                outs["steps"] = v.opt.step.item()
                # ... and the way the fixed-ratio loop issues them: learn_many(8) for V then learn_many(4) for a P-learner on its own
                # communicator (scripts/train_pql.py), twice with a hand-off in between -- the slot-graph / run-graph capture
                # warm-ups and the per-step collectives must stay rank-symmetric or this deadlocks / diverges
                from pql_amd.algo.pql_p_learner import PQLPLearner
                from pql_amd.utils.dp import component_groups
                groups = component_groups(dist.group.WORLD, ("p",))
                p = PQLPLearner((O,), A, make_cfg(False, B=B // world, graph=True), process_group=groups["p"])
                p.actor.load_state_dict(_sd(dd.mlp_state(O, A, 11)))
                p.use_private_rng(77 + rank)
                for it in range(2):
                    data = tuple(t.to(dev) for t in _fill(O, A, 100, 900 + 10 * it + rank))   # every rank inserts its OWN rows
                    critic, _, _ = v.update(p.actor, data, norm, 0)
                    p.update(critic, data[0], norm, 0)
                    v.learn_many(8)
                    p.learn_many(4)
                torch.cuda.synchronize()
                outs["many_v"] = v.critic.arena.data.cpu().numpy(); outs["many_p"] = p.actor.arena.data.cpu().numpy()
                outs["many_counts"] = (v.update_count, p.update_count)
        ret[rank] = outs
    finally:
        dist.destroy_process_group()


def test_data_parallel_two_ranks_match_single_gpu_step(golden, dev):
    """Two ranks share the one GPU of the test box (gloo rehearsal backend): sharded batch + all-reduce + replicated
    optimiser reproduces the golden single-GPU step, and the graph-captured DP step keeps the replicas bit-identical."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_dp_rank, args=(2, port, ret), nprocs=2, join=True)
    g = golden("learners")
    for r in (0, 1):
        np.testing.assert_allclose(ret[r]["injected"], g["v_s0_p_net_q1.net.2.weight"], rtol=5e-5, atol=5e-7)
        assert ret[r]["steps"] == 3
    # different seeds-per-rank draws differ, but parameters are replicated: identical after the all-reduced steps
    assert np.array_equal(ret[0]["graph"], ret[1]["graph"])
    assert np.array_equal(ret[0]["many_v"], ret[1]["many_v"]) and np.array_equal(ret[0]["many_p"], ret[1]["many_p"])
    assert ret[0]["many_counts"] == ret[1]["many_counts"] == (3 + 16, 8)


def _dp_bucket_rank(rank, port, ret):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)   # RCCL with one rank: the production collectives, nothing to sum
    try:
        from pql_amd.algo.pql_v_learner import PQLVLearner
        from pql_amd.models.mlp import TanhMLPPolicy
        from pql_amd.utils.dp import component_groups
        dev = torch.device("cuda:0")
        O, A, B = 24, 6, 1024
        pg = component_groups(dist.group.WORLD, ("v",))["v"]
        outs = {}
        for distl in (False, True):
            for name, buckets, graph, captured in (("one_eager", "one", False, "0"), ("layer_eager", "layer", False, "0"),
                                                   ("one_graph", "one", True, "0"), ("layer_graph", "layer", True, "0"),
                                                   ("layer_graph_captured_collectives", "layer", True, "1")):
                os.environ["PQL_DP_GRAPH_COLLECTIVE"] = captured
                cfg = make_cfg(distl, B=B, memory=5000, hidden=[256, 256, 128], graph=graph)
                cfg.algo.dp_buckets = buckets
                v = PQLVLearner((O,), A, cfg, process_group=pg)
                assert (v._buckets == [(3, 2), (1, 1), (0, 0)]) if buckets == "layer" else v._buckets is None
                K = 51 if distl else 1
                v.critic.load_state_dict(_sd(dd.doubleq_state(O, A, K, 21, hidden=[256, 256, 128])))
                v.critic_target.arena.data.copy_(v.critic.arena.data)
                actor = TanhMLPPolicy((O,), A, hidden_layers=[256, 256, 128]).to(dev)
                actor.load_state_dict(_sd(dd.mlp_state(O, A, 11, hidden=[256, 256, 128])))
                v.update(actor, tuple(t.to(dev) for t in _fill(O, A, 3000, 810)), None, 0)
                v.use_private_rng(99)
                for _ in range(4):
                    v.learn()
                torch.cuda.synchronize()
                outs[(distl, name)] = (v.critic.arena.data.cpu().numpy(), v._ws["grads"].cpu().numpy(), v.loss_ring.cpu().numpy(),
                                       int(v.opt.step.item()))
        ret[0] = outs
    finally:
        dist.destroy_process_group()


def test_gradient_buckets_leave_the_bits_of_the_single_collective(dev):
    """SURVEY 8(e), the all-reduce overlapped with backward: the critic's gradient all-reduced in per-layer buckets issued
    behind each layer's slab sum (`algo.dp_buckets=layer`: pqlk_mlp_backward_layers + one grouped RCCL launch per bucket) must
    leave the gradient, the parameters and the losses of the one-collective path bit for bit -- eager, as one hipGraph per
    bucket around eager collectives, and with the collectives captured; scalar and C51 heads.  One RCCL rank (the only
    RCCL this pool can run); the two-rank sum is covered with gloo by test_data_parallel_two_ranks_match_single_gpu_step."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_dp_bucket_rank, args=(port, ret), nprocs=1, join=True)
    outs = ret[0]
    for distl in (False, True):
        ref = outs[(distl, "one_eager")]
        assert ref[3] == 4 and np.isfinite(ref[0]).all() and np.abs(ref[1]).max() > 0
        for name in ("layer_eager", "one_graph", "layer_graph", "layer_graph_captured_collectives"):
            got = outs[(distl, name)]
            for a, b, what in zip(ref, got, ("parameters", "gradient", "loss ring", "step count")):
                assert np.array_equal(a, b), f"{name} (distl={distl}): {what} differ from the single-collective eager path"


# --------------------------------------------------------------------------- evaluator
@pytest.mark.parametrize("subprocess_mode", [False, True])
def test_evaluator_on_gpu_matches_plain_rollout(dev, tmp_path, subprocess_mode):
    """In-process (own HIP stream, cooperative polls) and subprocess (pipe protocol) evaluators both reproduce a plain
    rollout loop of the same actor on the same synthetic env, and leave a loadable best checkpoint."""
    from types import SimpleNamespace
    from pql_amd.envs.synthetic import create_task_env
    from pql_amd.models.mlp import DoubleQ, TanhMLPPolicy
    from pql_amd.utils.cfg import load_cfg
    from pql_amd.utils.common import Tracker
    from pql_amd.utils.evaluator import Evaluator
    from pql_amd.utils.model_util import load_model
    from pql_amd.utils.torch_util import RunningMeanStd
    cfg = load_cfg(["task.name=Toy", "task.episode_length=40", "eval_num_envs=150", "device=cuda:0", "eval_steps_per_poll=16",
                    f"eval_subprocess={subprocess_mode}"])
    torch.manual_seed(3)
    actor, critic = TanhMLPPolicy((8,), 2).to(dev), DoubleQ((8,), 2).to(dev)
    rms = RunningMeanStd(shape=(8,), device=dev)
    rms.update(torch.randn(512, 8, device=dev) * 2 + 0.5)
    # plain loop, host-side Tracker (the reference's form)
    n = 150
    env = create_task_env(cfg, num_envs=n)
    rt, lt = Tracker(n), Tracker(n)
    cr, cl = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    obs = env.reset()
    with torch.no_grad():
        for _ in range(env.max_episode_length):
            obs, rew, done, _ = env.step(actor(rms.normalize(obs)))
            cr += rew; cl += 1
            idx = torch.where(done)[0]
            rt.update(cr[idx]); lt.update(cl[idx])
            cr[idx] = 0; cl[idx] = 0
    ev = Evaluator(cfg, wandb_run=SimpleNamespace(dir=str(tmp_path)))
    ev.eval_policy(actor, critic, step=1, normalizer=rms)
    actor.arena.data.zero_()      # the evaluation must run on the snapshot taken at eval_policy time
    if not subprocess_mode:
        polls = 0
        while not ev.parent.poll():
            polls += 1
            assert polls < 100000
    got = ev.parent.recv()
    ev.close()
    assert abs(got["eval/return"] - rt.mean()) < 1e-5 and abs(got["eval/episode_length"] - lt.mean()) < 1e-5
    a2 = TanhMLPPolicy((8,), 2).to(dev)
    assert load_model(a2, "actor", str(tmp_path / "model.pth")) and float(a2.arena.data.abs().sum()) > 0
    holder = RunningMeanStd(shape=(8,), device=dev)
    assert load_model(holder, "obs_rms", str(tmp_path / "model.pth")) and torch.allclose(holder.mean, rms.mean)


def test_evaluator_against_the_oracle_composition(dev, tmp_path):
    """f2 against the ORACLE, not against another HIP loop: the evaluation (pql/utils/evaluator.py:41-121 -- un-clamped
    normalise -> actor -> env.step -> per-env return / length accumulators -> zero-filled windows of capacity
    eval_num_envs -> their means) recomputed on the CPU from the env transitions the product saw (the env is an input of
    both sides), with the oracle's actor forward and a deque.  Bars: every action within 1e-5 of the oracle's, result
    means within 1e-6 relative, and the evaluated weights are those of the snapshot taken at eval_policy time."""
    from collections import deque
    from types import SimpleNamespace
    from oracle import pql_ref_cpu as ref
    from pql_amd.envs.synthetic import create_task_env
    from pql_amd.models.mlp import DoubleQ, TanhMLPPolicy
    from pql_amd.utils.cfg import load_cfg
    from pql_amd.utils.evaluator import Evaluator
    from pql_amd.utils.torch_util import RunningMeanStd
    n, O, A = 150, 8, 2
    cfg = load_cfg(["task.name=Toy", "task.episode_length=40", f"eval_num_envs={n}", "device=cuda:0", "eval_steps_per_poll=16"])
    rec = {}

    class Recording:
        def __init__(self, env):
            self.env, self.log, self.first_obs = env, [], None
            self.observation_space, self.action_space = env.observation_space, env.action_space
            self.max_episode_length, self.num_envs = env.max_episode_length, env.num_envs

        def reset(self):
            self.first_obs = self.env.reset().clone()
            return self.first_obs

        def step(self, action):
            out = self.env.step(action)
            self.log.append((action.clone(), out[0].clone(), out[1].clone(), out[2].clone()))
            return out

    def make_env(c, num_envs=None):
        rec["env"] = Recording(create_task_env(c, num_envs=num_envs))
        return rec["env"]

    ast = dd.mlp_state(O, A, 17)
    actor = TanhMLPPolicy((O,), A).to(dev)
    actor.load_state_dict({k: T(v) for k, v in ast.items()})
    critic = DoubleQ((O,), A).to(dev)
    rms = RunningMeanStd(shape=(O,), device=dev)
    rms.update(T(dd.uniform((512, O), 5, -3, 4)).to(dev))
    ev = Evaluator(cfg, wandb_run=SimpleNamespace(dir=str(tmp_path)), create_task_env_func=make_env)
    ev.eval_policy(actor, critic, step=7, normalizer=rms)
    actor.arena.data.zero_()          # later training steps must not leak into the evaluation
    polls = 0
    while not ev.parent.poll():
        polls += 1
        assert polls < 100000
    got = ev.parent.recv()
    ev.close()
    env = rec["env"]
    assert len(env.log) == env.max_episode_length == 40
    # ---- oracle composition on the recorded transitions
    apar = ref.params_from_state(ast)
    mean, var, eps = (t.cpu() if torch.is_tensor(t) else t for t in rms.get_states())
    ret_win, len_win = deque([0.0] * n, maxlen=n), deque([0.0] * n, maxlen=n)     # Tracker(eval_num_envs): zero-filled
    cur_ret, cur_len = torch.zeros(n), torch.zeros(n)
    obs = env.first_obs.cpu()
    for logged_act, nobs, rew, done in env.log:
        act = ref.actor_forward_ref(apar, ref.normalize_ref(obs, (mean, var, eps), clamp=False))     # evaluator.py:66-68
        torch.testing.assert_close(logged_act.cpu(), act, rtol=0, atol=1e-5)
        cur_ret += rew.cpu(); cur_len += 1
        fin = done.cpu().bool()
        ret_win.extend(cur_ret[fin].tolist()); len_win.extend(cur_len[fin].tolist())
        cur_ret[fin] = 0; cur_len[fin] = 0
        obs = nobs.cpu()
    assert sum(x != 0 for x in len_win) > 20                       # episodes did finish (mean length 40 over 40 steps x 150 envs)
    assert got["eval/return"] == pytest.approx(float(np.mean(ret_win)), rel=1e-6, abs=1e-7)
    assert got["eval/episode_length"] == pytest.approx(float(np.mean(len_win)), rel=1e-6)


# --------------------------------------------------------------------------- SAC
def _sac_cfg(extra=()):
    from pql_amd.utils.cfg import load_cfg
    return load_cfg(["algo=sac_algo", "task.name=Toy", "num_envs=64", "algo.batch_size=64", "algo.memory_size=400",
                     "device=cuda:0", "sim_device=cuda:0", *extra])


def test_sac_golden_trace(golden, dev):
    """AgentSAC.update_once (HIP launch sequence) vs three iterations of the reference's AgentSAC.update_critic /
    update_actor / soft_update on identical samples and rsample draws: losses, log_alpha, every parameter tensor."""
    from pql_amd.algo.sac import AgentSAC
    from pql_amd.envs.synthetic import create_task_env
    from pql_amd.replay.simple_replay import ReplayBuffer
    g = golden("sac"); O, A = 8, 2
    cfg = _sac_cfg()
    agent = AgentSAC(create_task_env(cfg), cfg)
    agent.actor.load_state_dict(_sd(dd.mlp_state(O, 2 * A, 11)))
    agent.critic.load_state_dict(_sd(dd.doubleq_state(O, A, 1, 21))); agent.critic_target.arena.data.copy_(agent.critic.arena.data)
    agent.obs_rms.mean, agent.obs_rms.var = T(g["sac_norm_mean"]).to(dev), T(g["sac_norm_var"]).to(dev)
    memory = ReplayBuffer(400, (O,), A, device=dev)
    memory.add_to_buffer(tuple(t.to(dev) for t in _fill(O, A, 300, 810)))
    for s in range(3):
        agent.update_once(memory, indices=T(g["sac_idx"][s]), eps_next=T(g["sac_eps"][2 * s]), eps_cur=T(g["sac_eps"][2 * s + 1]))
        np.testing.assert_allclose(agent.closs[s % 5].item(), g["sac_closs"][s], rtol=2e-5)
        np.testing.assert_allclose(agent.aloss[s % 5].item(), g["sac_aloss"][s], rtol=2e-5)
        np.testing.assert_allclose(agent.log_alpha.item(), g["sac_log_alpha"][s], rtol=1e-5)
        _check_module(agent.actor, g, f"sac_s{s}_a_")
        _check_module(agent.critic, g, f"sac_s{s}_c_")
        _check_module(agent.critic_target, g, f"sac_s{s}_t_")
    np.testing.assert_allclose(agent.actor.layout.weight(agent.actor.arena.data, 0, 3).cpu().numpy(), g["sac_final_actor_last_w"],
                               rtol=5e-5, atol=5e-7)
    np.testing.assert_allclose(agent.critic.layout.weight(agent.critic.arena.data, 0, 3).cpu().numpy(), g["sac_final_q1_last_w"],
                               rtol=5e-5, atol=5e-7)
    assert abs(agent.get_alpha(scalar=True) - np.exp(g["sac_log_alpha"][2])) < 1e-6


def test_sac_update_vs_oracle_allegro_shape(dev):
    """AllegroHand shapes at batch 4096 with the BASELINE hidden sizes, two iterations vs the oracle's SACRef (learned and fixed
    temperature)."""
    from oracle import pql_ref_cpu as ref
    from pql_amd.algo.sac import AgentSAC
    from pql_amd.envs.synthetic import create_task_env
    from pql_amd.replay.simple_replay import ReplayBuffer
    O, A, B, rows = 88, 16, 4096, 6000
    for fixed in (None, 0.2):
        cfg = _sac_cfg(["task.name=AllegroHand", f"algo.batch_size={B}", "algo.memory_size=8000"] + ([f"algo.alpha={fixed}"] if fixed else []))
        cfg.algo.hidden_layers = [512, 512, 256]
        agent = AgentSAC(create_task_env(cfg), cfg)
        ast, cst = dd.mlp_state(O, 2 * A, 11, hidden=(512, 512, 256)), dd.doubleq_state(O, A, 1, 21, hidden=(512, 512, 256))
        agent.actor.load_state_dict(_sd(ast))
        agent.critic.load_state_dict(_sd(cst)); agent.critic_target.arena.data.copy_(agent.critic.arena.data)
        mean, var = T(dd.uniform((O,), 801, -0.5, 0.5)), T(dd.uniform((O,), 802, 0.5, 2.0))
        agent.obs_rms.mean, agent.obs_rms.var = mean.to(dev), var.to(dev)
        data = _fill(O, A, rows, 77)
        memory = ReplayBuffer(8000, (O,), A, device=dev)
        memory.add_to_buffer(tuple(t.to(dev) for t in data))
        orc = ref.SACRef(O, A, ref.HyperRef(batch_size=B), 8000, ref.params_from_state(ast), ref.params_from_state(cst, "net_q1.net."),
                         ref.params_from_state(cst, "net_q2.net."), alpha_lr=cfg.algo.alpha_lr, alpha=fixed)
        orc.ring.insert(*data); orc.norm = (mean, var, 1e-4)
        for s in range(2):
            idx = T(dd.integers((B,), 40 + s, rows))
            e1, e2 = T(dd.uniform((B, A), 50 + s, -2, 2)), T(dd.uniform((B, A), 60 + s, -2, 2))
            cl, al_, _ = orc.update_once(idx, e1, e2)
            agent.update_once(memory, indices=idx, eps_next=e1, eps_cur=e2)
            np.testing.assert_allclose(agent.closs[s % 5].item(), cl, rtol=5e-5)
            np.testing.assert_allclose(agent.aloss[s % 5].item(), al_, rtol=5e-5, atol=1e-6)
        if fixed is None:
            np.testing.assert_allclose(agent.log_alpha.item(), float(orc.log_alpha.detach()), rtol=1e-5)
        else:
            assert abs(agent.get_alpha(scalar=True) - fixed) < 1e-7
        lay = agent.critic.layout
        for n, net in enumerate((orc.q1, orc.q2)):
            for l in range(lay.n_layers):
                np.testing.assert_allclose(lay.weight(agent.critic.arena.data, n, l).cpu().numpy(), net[2 * l].detach().numpy(),
                                           rtol=1e-5, atol=1e-5)
        for l in range(agent.actor.layout.n_layers):
            np.testing.assert_allclose(agent.actor.layout.weight(agent.actor.arena.data, 0, l).cpu().numpy(),
                                       orc.actor[2 * l].detach().numpy(), rtol=1e-5, atol=1e-5)


def test_train_baselines_entry_point_sac(dev):
    import importlib.util, os
    from pql_amd.utils.cfg import load_cfg
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("train_baselines", os.path.join(root, "scripts", "train_baselines.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    out = mod.main(load_cfg(["algo=sac_algo", "task.name=Toy", "num_envs=64", "algo.batch_size=256", "algo.memory_size=100000",
                             "device=cuda:0", "sim_device=cuda:0", "max_step=4000", "algo.update_times=4"]))
    assert out["global_steps"] > 4000 and np.isfinite(out["train/critic_loss"]) and np.isfinite(out["train/actor_loss"])
    assert 0 < out["train/alpha"] < 1.0   # entropy above target at the start: the temperature decreases from 1


# --------------------------------------------------------------------------- CrossQ
def _crossq_cfg(extra=()):
    from pql_amd.utils.cfg import load_cfg
    return load_cfg(["algo=crossq_algo", "task.name=Toy", "num_envs=64", "algo.batch_size=64", "algo.memory_size=400",
                     "device=cuda:0", "sim_device=cuda:0", *extra])


def _check_bn_critic(critic, g, prefix, skip_pre_norm_bias=True):
    for key, view in critic.named_views():
        layer = int(key.split(".")[2])
        if skip_pre_norm_bias and key.endswith(".bias") and layer % 3 == 0 and layer < 9:
            continue   # zero-gradient parameter moved by Adam on rounding noise: see tests/test_oracle_golden.py::test_crossq_trace
        np.testing.assert_allclose(dd.summarize(view.cpu().numpy()), g[f"{prefix}{key}"], rtol=5e-5, atol=1e-5, err_msg=prefix + key)


def test_batchnorm_critic_forward_golden(golden, dev):
    """DoubleQBatchNorm (one-layer GEMM calls + pqlk_batch_moments + pqlk_bn_elu_forward) vs the reference module: training
    mode on a 128-row batch, the running statistics it leaves, then eval mode."""
    from pql_amd.models.batchnorm import DoubleQBatchNorm
    g = golden("crossq"); O, A, B = 8, 2, 64
    q = DoubleQBatchNorm((O,), A).to(dev)
    q.load_state_dict(_sd(dd.bn_critic_state(O, A, 41)), strict=False)
    xk, ak = T(dd.uniform((2 * B, O), 71, -2, 2)).to(dev), T(dd.uniform((2 * B, A), 72, -1, 1)).to(dev)
    q.train()
    q1, q2 = q.get_q1_q2(xk, ak)
    np.testing.assert_allclose(q1.cpu().numpy(), g["cq_kat_train_q1"], atol=5e-6); np.testing.assert_allclose(q2.cpu().numpy(), g["cq_kat_train_q2"], atol=5e-6)
    sd = q.state_dict()
    np.testing.assert_allclose(sd["net_q1.net.1.running_mean"].cpu().numpy(), g["cq_kat_running_mean_l0"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(sd["net_q1.net.1.running_var"].cpu().numpy(), g["cq_kat_running_var_l0"], rtol=1e-5)
    assert int(sd["net_q2.net.7.num_batches_tracked"]) == 1
    q.eval()
    q1, q2 = q.get_q1_q2(xk, ak)
    np.testing.assert_allclose(q1.cpu().numpy(), g["cq_kat_eval_q1"], atol=5e-6); np.testing.assert_allclose(q2.cpu().numpy(), g["cq_kat_eval_q2"], atol=5e-6)
    assert int(q.state_dict()["net_q2.net.7.num_batches_tracked"]) == 1            # eval does not touch the statistics
    np.testing.assert_allclose(q.get_q_min(xk, ak).cpu().numpy(), np.minimum(g["cq_kat_eval_q1"], g["cq_kat_eval_q2"]), atol=5e-6)


def test_crossq_golden_trace(golden, dev):
    """AgentCrossQ.update_once (HIP) vs three iterations of the reference's AgentCrossQ.update_critic / update_actor: losses,
    actor, Linear weights, BatchNorm gamma / beta and running statistics."""
    from pql_amd.algo.crossq import AgentCrossQ
    from pql_amd.envs.synthetic import create_task_env
    from pql_amd.replay.simple_replay import ReplayBuffer
    g = golden("crossq"); O, A = 8, 2
    cfg = _crossq_cfg()
    agent = AgentCrossQ(create_task_env(cfg), cfg)
    agent.actor.load_state_dict(_sd(dd.mlp_state(O, A, 11)))
    agent.critic.load_state_dict(_sd(dd.bn_critic_state(O, A, 41)), strict=False)
    agent.obs_rms.mean, agent.obs_rms.var = T(g["cq_norm_mean"]).to(dev), T(g["cq_norm_var"]).to(dev)
    memory = ReplayBuffer(400, (O,), A, device=dev)
    memory.add_to_buffer(tuple(t.to(dev) for t in _fill(O, A, 300, 810)))
    for s in range(3):
        agent.update_once(memory, indices=T(g["cq_idx"][s]), noise=T(g["cq_noise"][s]))
        np.testing.assert_allclose(agent.closs[s % 5].item(), g["cq_closs"][s], rtol=2e-5)
        np.testing.assert_allclose(agent.aloss[s % 5].item(), g["cq_aloss"][s], rtol=2e-5)
        _check_module(agent.actor, g, f"cq_s{s}_a_")
        _check_bn_critic(agent.critic, g, f"cq_s{s}_c_")
        sd = agent.critic.state_dict()
        for k in sd:
            if k.endswith("running_var"):
                np.testing.assert_allclose(sd[k].cpu().numpy(), g[f"cq_s{s}_r_{k}"], rtol=2e-5, atol=2e-6, err_msg=k)
            if k.endswith("running_mean"):   # contains the incomparable pre-norm bias (see the oracle test)
                np.testing.assert_allclose(sd[k].cpu().numpy(), g[f"cq_s{s}_r_{k}"], rtol=2e-5, atol=5e-4, err_msg=k)
    np.testing.assert_allclose(agent.actor.layout.weight(agent.actor.arena.data, 0, 3).cpu().numpy(), g["cq_final_actor_last_w"],
                               rtol=5e-5, atol=5e-7)
    np.testing.assert_allclose(agent.critic.weight(0, 3).cpu().numpy(), g["cq_final_q1_last_w"], rtol=5e-5, atol=1e-5)
    np.testing.assert_allclose(agent.critic.bn_param(0, 0, "gamma").cpu().numpy(), g["cq_final_q1_bn0_gamma"], rtol=5e-5, atol=1e-5)


def test_crossq_update_vs_oracle_allegro_shape(dev):
    """AllegroHand shapes, batch 2048 (joint critic batch 4096), default hidden sizes: two iterations vs the oracle's CrossQRef."""
    from oracle import pql_ref_cpu as ref
    from pql_amd.algo.crossq import AgentCrossQ
    from pql_amd.envs.synthetic import create_task_env
    from pql_amd.replay.simple_replay import ReplayBuffer
    O, A, B, rows = 88, 16, 2048, 5000
    cfg = _crossq_cfg(["task.name=AllegroHand", f"algo.batch_size={B}", "algo.memory_size=8000"])
    agent = AgentCrossQ(create_task_env(cfg), cfg)
    ast, cst = dd.mlp_state(O, A, 11), dd.bn_critic_state(O, A, 41)
    agent.actor.load_state_dict(_sd(ast)); agent.critic.load_state_dict(_sd(cst), strict=False)
    mean, var = T(dd.uniform((O,), 801, -0.5, 0.5)), T(dd.uniform((O,), 802, 0.5, 2.0))
    agent.obs_rms.mean, agent.obs_rms.var = mean.to(dev), var.to(dev)
    data = _fill(O, A, rows, 77)
    memory = ReplayBuffer(8000, (O,), A, device=dev)
    memory.add_to_buffer(tuple(t.to(dev) for t in data))
    lin, bn = [], []
    for pre in ("net_q1.net.", "net_q2.net."):
        lin.append([T(cst[f"{pre}{3 * l}.{k}"]) for l in range(4) for k in ("weight", "bias")])
        bn.append([T(cst[f"{pre}{3 * l + 1}.{k}"]) for l in range(3) for k in ("weight", "bias")])
    orc = ref.CrossQRef(O, A, ref.HyperRef(batch_size=B), 8000, ref.params_from_state(ast), lin, bn)
    orc.ring.insert(*data); orc.norm = (mean, var, 1e-4)
    for s in range(2):
        idx, draw = T(dd.integers((B,), 40 + s, rows)), T(dd.uniform((B, A), 50 + s, -2, 2))
        cl, al_ = orc.update_once(idx, draw)
        agent.update_once(memory, indices=idx, noise=draw)
        np.testing.assert_allclose(agent.closs[s % 5].item(), cl, rtol=5e-5)
        np.testing.assert_allclose(agent.aloss[s % 5].item(), al_, rtol=5e-5, atol=1e-6)
    for n in range(2):
        for l in range(4):
            np.testing.assert_allclose(agent.critic.weight(n, l).cpu().numpy(), orc.q_lin[n][2 * l].detach().numpy(), rtol=1e-5, atol=1e-5)
            if l < 3:
                np.testing.assert_allclose(agent.critic.bn_param(n, l, "gamma").cpu().numpy(), orc.q_bn[n][2 * l].detach().numpy(), rtol=1e-5, atol=1e-5)
                np.testing.assert_allclose(agent.critic.bn_param(n, l, "beta").cpu().numpy(), orc.q_bn[n][2 * l + 1].detach().numpy(), rtol=1e-5, atol=1e-5)
                np.testing.assert_allclose(agent.critic.running(n, l, "var").cpu().numpy(), orc.q_stats[n][2 * l + 1].numpy(), rtol=2e-5, atol=2e-6)
    for l in range(agent.actor.layout.n_layers):
        np.testing.assert_allclose(agent.actor.layout.weight(agent.actor.arena.data, 0, l).cpu().numpy(),
                                   orc.actor[2 * l].detach().numpy(), rtol=1e-5, atol=1e-5)


def test_train_baselines_entry_point_crossq(dev):
    import importlib.util, os
    from pql_amd.utils.cfg import load_cfg
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("train_baselines", os.path.join(root, "scripts", "train_baselines.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    out = mod.main(load_cfg(["algo=crossq_algo", "task.name=Toy", "num_envs=64", "algo.batch_size=256", "algo.memory_size=100000",
                             "device=cuda:0", "sim_device=cuda:0", "max_step=4000", "algo.update_times=4"]))
    assert out["global_steps"] > 4000 and np.isfinite(out["train/critic_loss"]) and np.isfinite(out["train/actor_loss"])
