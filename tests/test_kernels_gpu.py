"""Parity of the HIP kernels (through the C ABI / ctypes) against the CPU oracle and the committed
golden vectors.  Needs an MI355X: run with `pytest -m gpu`.

Bars: bit-exact for byte/index work (ring, gather, normalise, n-step); fp32 MLP outputs within 1e-5 of
the reference (north_star); gradients / optimiser within the rtol/atol written at each assert."""
import ctypes as C

import numpy as np
import pytest
import torch

import detdata as dd

pytestmark = pytest.mark.gpu

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ref():
    from oracle import pql_ref_cpu
    return pql_ref_cpu


# --------------------------------------------------------------------------- ring
@pytest.mark.parametrize("name", ["wrap4", "exact", "ragged"])
def test_ring_insert_and_sample(golden, dev, name):
    from pql_amd.replay.simple_replay import ReplayBuffer
    g = golden("replay")
    cap, O, A = (int(v) for v in g[f"ring_{name}_meta"])
    rb = ReplayBuffer(cap, (O,), A, device=dev)
    for step, m in enumerate(g[f"ring_{name}_inserts"]):
        seed = 100 + step
        traj = (T(dd.uniform((m, O), seed)), T(dd.uniform((m, A), seed + 20)), T(dd.uniform((m, 1), seed + 40)),
                T(dd.uniform((m, O), seed + 60)), T(dd.bernoulli((m, 1), seed + 80, 0.3)))
        rb.add_to_buffer(tuple(t.to(dev) for t in traj))
        assert [rb.next_p, rb.cur_capacity, int(rb.if_full)] == g[f"ring_{name}_trace"][step].tolist()
    for nm, t in (("obs", rb.buf_obs), ("act", rb.buf_action), ("rew", rb.buf_reward), ("nobs", rb.buf_next_obs),
                  ("done", rb.buf_done)):
        assert np.array_equal(t.cpu().numpy(), g[f"ring_{name}_{nm}"]), nm
    out = rb.sample_batch(16, device=dev, indices=T(g[f"ring_{name}_idx"]))
    for nm, t in zip(("s_obs", "s_act", "s_rew", "s_nobs", "s_done"), out):
        assert t.dtype == torch.float32 and t.is_cuda
        assert np.array_equal(t.cpu().numpy(), g[f"ring_{name}_{nm}"]), nm
    # un-injected: exactly one int64 randint draw of shape (B,) on the ring's device (Appendix B)
    torch.manual_seed(7)
    expect = torch.randint(rb.cur_capacity, size=(16,), device=dev)
    torch.manual_seed(7)
    assert torch.equal(rb.draw_indices(16), expect)


@pytest.mark.parametrize("seed", range(8))
def test_ring_random_insert_sequences_vs_oracle(dev, ref, seed):
    """Seeded random rings (capacity, widths) and insert sequences -- single rows, exact fills, inserts of the whole capacity,
    several wraps -- against the oracle's ring: pointer state after every insert, the five field views and a gather that names every
    slot, all bit for bit; the P-learner's obs-only ring runs the same sequence."""
    from pql_amd import _lib as L
    from pql_amd.replay.simple_replay import RecordRing, ReplayBuffer, ring_plan
    rs = np.random.RandomState(1000 + seed)
    cap = int(rs.choice([1, 2, 7, 32, 33, 100, 257]))
    O, A = int(rs.choice([1, 3, 8, 13, 33])), int(rs.choice([1, 2, 5, 17]))
    rb, oracle = ReplayBuffer(cap, (O,), A, device=dev), ref.RingRef(cap, O, A)
    pr, poracle = RecordRing(cap, O, -1, dev), ref.ObsRingRef(cap, O)   # (the P-learner keeps the pointer state itself)
    p_next, p_full, p_cur = 0, False, 0
    for step in range(14):
        m = int(rs.choice([1, cap, max(1, cap // 2), int(rs.randint(1, cap + 1)), max(1, cap - 1)]))
        sd = 5000 + 100 * seed + step
        traj = (T(dd.uniform((m, O), sd)), T(dd.uniform((m, A), sd + 20)), T(dd.uniform((m, 1), sd + 40)), T(dd.uniform((m, O), sd + 60)),
                T(dd.bernoulli((m, 1), sd + 80, 0.3)))
        oracle.insert(*traj); poracle.insert(traj[0])
        rb.add_to_buffer(tuple(t.to(dev) for t in traj))
        segs, p_next, p_full, p_cur = ring_plan(p_next, p_full, cap, m)
        pr.insert_segments(segs, traj[0].to(dev))
        assert (rb.next_p, rb.cur_capacity, rb.if_full) == (oracle.next_p, oracle.cur_capacity, oracle.if_full), step
        assert (p_next, p_cur, p_full) == (poracle.next_p, poracle.cur_capacity, poracle.if_full), step
    for mine, want in ((rb.buf_obs, oracle.obs), (rb.buf_action, oracle.act), (rb.buf_reward, oracle.rew), (rb.buf_next_obs, oracle.nobs),
                       (rb.buf_done, oracle.done)):
        assert np.array_equal(mine.cpu().numpy(), want.numpy())
    idx = torch.from_numpy(np.concatenate([np.arange(cap), rs.randint(0, cap, size=50)])).to(torch.int64)
    for mine, want in zip(rb.sample_batch(idx.numel(), device=dev, indices=idx), oracle.gather(idx)):
        assert mine.dtype == torch.float32 and np.array_equal(mine.cpu().numpy(), want.numpy())
    x_o = torch.full((idx.numel(), L.ld(O)), 7.0, device=dev)
    idx_d = idx.to(dev)
    L.check(L.lib.pqlk_replay_gather_fused(C.byref(pr.desc), L.ptr(idx_d), idx.numel(), None, None, 0.0, 1, None, 0, None, L.ptr(x_o),
                                           x_o.stride(0), None, None, L.stream(dev)))
    assert np.array_equal(x_o[:, :O].cpu().numpy(), poracle.gather(idx).numpy()) and torch.all(x_o[:, O:] == 0)


def test_ring_rejects_cpu_and_oversize(dev):
    from pql_amd._lib import PqlkError
    from pql_amd.replay.simple_replay import ReplayBuffer
    with pytest.raises(PqlkError):
        ReplayBuffer(10, (3,), 2, device="cpu")
    rb = ReplayBuffer(10, (3,), 2, device=dev)
    z = lambda *s: torch.zeros(*s, device=dev)  # noqa: E731
    with pytest.raises(RuntimeError):
        rb.add_to_buffer((z(25, 3), z(25, 2), z(25, 1), z(25, 3), z(25, 1)))


@pytest.mark.parametrize("O,A", [(8, 2), (88, 16), (211, 20), (108, 21)])
def test_gather_fused_bit_exact(dev, ref, O, A):
    """sample + normalize(+-5 clamp) + cat, vs oracle; ragged O/A exercise the unaligned field paths."""
    from pql_amd import _lib as L
    from pql_amd.replay.simple_replay import ReplayBuffer
    cap, B = 500, 77
    rb = ReplayBuffer(cap, (O,), A, device=dev)
    ring = ref.RingRef(cap, O, A)
    data = (T(dd.uniform((cap, O), 1, -30, 30)), T(dd.uniform((cap, A), 2)), T(dd.uniform((cap, 1), 3)),
            T(dd.uniform((cap, O), 4, -30, 30)), T(dd.bernoulli((cap, 1), 5, 0.3)))
    rb.add_to_buffer(tuple(t.to(dev) for t in data)); ring.insert(*data)
    idx = T(dd.integers((B,), 6, cap))
    mean, var = T(dd.uniform((O,), 7)), T(dd.uniform((O,), 8, 0.01, 4.0))
    o, a, r, no, d = ring.gather(idx)
    on, non = ref.normalize_ref(o, (mean, var, 1e-4)), ref.normalize_ref(no, (mean, var, 1e-4))
    ld_sa, ld_o = L.ld(O + A), L.ld(O)
    f = dict(dtype=torch.float32, device=dev)
    x_sa = torch.full((B, ld_sa), 9.0, **f); xn_sa = torch.full((B, ld_sa), 9.0, **f); xn_o = torch.full((B, ld_o), 9.0, **f)
    rew = torch.empty(B, **f); done = torch.empty(B, **f)
    idx_d, mean_d, var_d = idx.to(dev), mean.to(dev), var.to(dev)   # keep device inputs alive across the async launch
    L.check(L.lib.pqlk_replay_gather_fused(C.byref(rb.ring.desc), L.ptr(idx_d), B, L.ptr(mean_d),
                                           L.ptr(var_d), 1e-4, 1, L.ptr(x_sa), ld_sa, L.ptr(xn_sa), L.ptr(xn_o), ld_o,
                                           L.ptr(rew), L.ptr(done), L.stream(dev)))
    x_sa, xn_sa, xn_o = x_sa.cpu(), xn_sa.cpu(), xn_o.cpu()
    assert torch.equal(x_sa[:, :O], on) and torch.equal(x_sa[:, O:O + A], a)
    assert torch.all(x_sa[:, O + A:] == 0)
    assert torch.equal(xn_sa[:, :O], non) and torch.all(xn_sa[:, O:O + A] == 9.0) and torch.all(xn_sa[:, O + A:] == 0)
    assert torch.equal(xn_o[:, :O], non) and torch.all(xn_o[:, O:] == 0)
    assert torch.equal(rew.cpu(), r.view(-1)) and torch.equal(done.cpu(), d.view(-1))


@pytest.mark.parametrize("O,A", [(88, 16), (211, 20), (108, 21)])
def test_gather_pads_zero_flag_and_multi_trip_grid(dev, ref, O, A):
    """PQLK_GATHER_PADS_ZERO (flag bit 1): the caller's tiles are zero-initialised, so the kernel must leave the pad columns
    alone and still produce the same data columns; 5000 rows also exercise the grid-stride trips and the ragged last trip."""
    from pql_amd import _lib as L
    from pql_amd.replay.simple_replay import ReplayBuffer
    cap, B = 3000, 5000
    rb = ReplayBuffer(cap, (O,), A, device=dev)
    ring = ref.RingRef(cap, O, A)
    data = (T(dd.uniform((cap, O), 11, -30, 30)), T(dd.uniform((cap, A), 12)), T(dd.uniform((cap, 1), 13)),
            T(dd.uniform((cap, O), 14, -30, 30)), T(dd.bernoulli((cap, 1), 15, 0.3)))
    rb.add_to_buffer(tuple(t.to(dev) for t in data)); ring.insert(*data)
    idx = T(dd.integers((B,), 16, cap))
    mean, var = T(dd.uniform((O,), 17)), T(dd.uniform((O,), 18, 0.01, 4.0))
    mean[3] = 0.0; var[5] = 0.0                       # a column whose sd is sqrt(eps) exactly
    o, a, r, no, d = ring.gather(idx)
    on, non = ref.normalize_ref(o, (mean, var, 1e-4)), ref.normalize_ref(no, (mean, var, 1e-4))
    ld_sa, ld_o = L.ld(O + A), L.ld(O)
    f = dict(dtype=torch.float32, device=dev)
    idx_d, mean_d, var_d = idx.to(dev), mean.to(dev), var.to(dev)
    for flags, pad_value in ((1, 0.0), (3, 9.0)):   # without the flag pads are zeroed, with it they are not touched
        x_sa = torch.full((B, ld_sa), 9.0, **f); xn_sa = torch.full((B, ld_sa), 9.0, **f)
        rew = torch.empty(B, **f); done = torch.empty(B, **f)
        L.check(L.lib.pqlk_replay_gather_fused(C.byref(rb.ring.desc), L.ptr(idx_d), B, L.ptr(mean_d), L.ptr(var_d), 1e-4, flags,
                                               L.ptr(x_sa), ld_sa, L.ptr(xn_sa), None, ld_o, L.ptr(rew), L.ptr(done), L.stream(dev)))
        xs, xns = x_sa.cpu(), xn_sa.cpu()
        assert torch.equal(xs[:, :O], on) and torch.equal(xs[:, O:O + A], a) and torch.all(xs[:, O + A:] == pad_value)
        assert torch.equal(xns[:, :O], non) and torch.all(xns[:, O:O + A] == 9.0) and torch.all(xns[:, O + A:] == pad_value)
        assert torch.equal(rew.cpu(), r.view(-1)) and torch.equal(done.cpu(), d.view(-1))


@pytest.mark.parametrize("obs_only", [False, True])
def test_gather_division_is_the_ieee_quotient(dev, obs_only):
    """The gather kernels normalise with ONE double-precision product per element (replay.hip: norm_div) instead of the fp32
    division sequence; it must be the IEEE quotient bit for bit -- checked here on 35 M arbitrary bit patterns (huge, denormal,
    signed zeros, infinities, NaN) against torch's fp32 division on the same device and numpy's on the host, for standard deviations
    spread over 12 decades and packed around 1."""
    from pql_amd import _lib as L
    from pql_amd.replay.simple_replay import RecordRing, ReplayBuffer
    O, A, cap = 88, 16, 200_000
    g = torch.Generator(device=dev).manual_seed(17)
    bits = torch.randint(-2 ** 31, 2 ** 31, (cap, O), device=dev, generator=g, dtype=torch.int64).to(torch.int32)
    obs = bits.view(torch.float32).clone()
    obs[::5] = torch.randn((obs[::5].shape), device=dev, generator=g) * 3                 # ordinary observations
    obs[1::97, :8] = torch.tensor([0.0, -0.0, float("inf"), -float("inf"), float("nan"), 1e-45, -1e-45, 3.4e38], device=dev)
    nobs = obs.flip(0).contiguous()
    if obs_only:
        ring = RecordRing(cap, O, -1, dev)
        ring.insert_segments([(0, 0, cap)], obs)
    else:
        rb = ReplayBuffer(cap, (O,), A, device=dev)
        rb.add_to_buffer((obs, torch.zeros((cap, A), device=dev), torch.zeros((cap, 1), device=dev), nobs, torch.zeros((cap, 1), device=dev)))
        ring = rb.ring
    assert torch.equal(ring.records[:, :O].view(torch.int32), obs.view(torch.int32))   # the ring holds the bit patterns untouched
    mean = torch.zeros(O, device=dev)                      # x - 0 = x exactly (and -0 - 0 = -0): the division sees the raw patterns
    sd_exp = torch.linspace(-6, 6, O, device=dev)
    var = (10.0 ** sd_exp) ** 2
    var[::3] = (1.0 + torch.rand(var[::3].shape, device=dev, generator=g)) ** 2
    eps = 1e-4
    sd = torch.sqrt(var + eps)
    idx = torch.randperm(cap, device=dev, generator=g)
    ld_sa, ld_o = L.ld(O + max(A, 0)), L.ld(O)
    outs = {}
    for name, flags in (("product", 0),):
        x_sa = torch.zeros((cap, ld_sa), device=dev); xn_sa = torch.zeros((cap, ld_sa), device=dev); x_o = torch.zeros((cap, ld_o), device=dev)
        L.check(L.lib.pqlk_replay_gather_fused(C.byref(ring.desc), L.ptr(idx), cap, L.ptr(mean), L.ptr(var), eps, flags, L.ptr(x_sa), ld_sa,
                                               None if obs_only else L.ptr(xn_sa), L.ptr(x_o), ld_o, None, None, L.stream(dev)))
        outs[name] = (x_sa[:, :O].clone(), (x_o if obs_only else xn_sa)[:, :O].clone())
    want = ((obs - mean) / sd)[idx], ((obs if obs_only else nobs) / sd)[idx]

    def same_bits(a, b):
        both_nan = torch.isnan(a) & torch.isnan(b)
        return bool(torch.all((a.view(torch.int32) == b.view(torch.int32)) | both_nan))
    for k in (0, 1):
        assert same_bits(outs["product"][k], want[k]), ("product vs torch division", k)
    with np.errstate(all="ignore"):
        host = (obs[idx][:4096].cpu().numpy() / sd.cpu().numpy()).astype(np.float32)   # numpy on the host: IEEE by construction
    got = outs["product"][0][:4096].cpu().numpy()
    assert np.array_equal(got.view(np.uint32)[~np.isnan(host)], host.view(np.uint32)[~np.isnan(host)]) and np.isnan(got[np.isnan(host)]).all()


def test_obs_ring_gather(dev, ref):
    from pql_amd import _lib as L
    from pql_amd.replay.simple_replay import RecordRing, ring_plan
    O, cap, B = 13, 40, 33
    ring = RecordRing(cap, O, -1, dev)
    oracle = ref.ObsRingRef(cap, O)
    next_p, full = 0, False
    for step, m in enumerate([17, 17, 17]):
        obs = T(dd.uniform((m, O), 40 + step))
        segs, next_p, full, cur = ring_plan(next_p, full, cap, m)
        ring.insert_segments(segs, obs.to(dev))
        oracle.insert(obs)
    assert (next_p, cur) == (oracle.next_p, oracle.cur_capacity)
    assert torch.equal(ring.records[:, :O].cpu(), oracle.mem)
    idx = T(dd.integers((B,), 9, cur))
    x_sa = torch.full((B, L.ld(O + 3)), 7.0, device=dev); x_o = torch.full((B, L.ld(O)), 7.0, device=dev)
    idx_d = idx.to(dev)
    L.check(L.lib.pqlk_replay_gather_fused(C.byref(ring.desc), L.ptr(idx_d), B, None, None, 0.0, 1, L.ptr(x_sa),
                                           x_sa.stride(0), None, L.ptr(x_o), x_o.stride(0), None, None, L.stream(dev)))
    assert torch.equal(x_o[:, :O].cpu(), oracle.gather(idx)) and torch.all(x_o[:, O:] == 0)
    assert torch.equal(x_sa[:, :O].cpu(), oracle.gather(idx)) and torch.all(x_sa[:, O:] == 0)


@pytest.mark.parametrize("O,B", [(88, 8192 + 5), (88, 4 * 8192), (8, 777), (63, 1001), (211, 2049), (256, 515), (260, 300)])
def test_obs_ring_gather_shapes_vs_oracle(dev, ref, O, B):
    """The obs-only ring's gather (several records per wave instruction up to O = 256, the generic kernel beyond) against the
    oracle's `memory[idx]` + normalize (pql_p_learner.py:49-52, common.py:139-145): bit-exact, pads zeroed or left alone as the
    flag word says, whatever launch shape the row count selects (one trip, several, a ragged last one)."""
    from pql_amd import _lib as L
    from pql_amd.replay.simple_replay import RecordRing
    cap = 5000
    ring = RecordRing(cap, O, -1, dev)
    mem = T(dd.uniform((cap, O), 70 + O, -3, 3))
    ring.insert_segments([(0, 0, cap)], mem.to(dev))
    idx = T(dd.integers((B,), 9 + O, cap))
    mean, var = T(dd.uniform((O,), 6, -0.5, 0.5)), T(dd.uniform((O,), 7, 0.5, 2.0))
    want_raw = mem[idx]
    want_norm = ref.normalize_ref(want_raw, (mean, var, 1e-4))
    idx_d, mean_d, var_d = idx.to(dev), mean.to(dev), var.to(dev)
    for norm in (False, True):
        for flags in (1, 1 | 2, 1 | 2 | (2 << 8) | (8 << 12), 1 | (4 << 8) | (4 << 12)):   # pads written / left; forced launch shapes
            x_sa = torch.full((B, L.ld(O + 5)), 7.0, device=dev); x_o = torch.full((B, L.ld(O)), 7.0, device=dev)
            L.check(L.lib.pqlk_replay_gather_fused(C.byref(ring.desc), L.ptr(idx_d), B, L.ptr(mean_d) if norm else None,
                                                   L.ptr(var_d) if norm else None, 1e-4, flags, L.ptr(x_sa), x_sa.stride(0), None,
                                                   L.ptr(x_o), x_o.stride(0), None, None, L.stream(dev)))
            want = want_norm if norm else want_raw
            pad = 7.0 if flags & 2 else 0.0
            assert torch.equal(x_o[:, :O].cpu(), want) and torch.all(x_o[:, O:] == pad), (norm, flags)
            assert torch.equal(x_sa[:, :O].cpu(), want) and torch.all(x_sa[:, O:] == pad), (norm, flags)
    # one destination only
    x_o = torch.zeros((B, L.ld(O)), device=dev)
    L.check(L.lib.pqlk_replay_gather_fused(C.byref(ring.desc), L.ptr(idx_d), B, None, None, 0.0, 3, None, 0, None, L.ptr(x_o), x_o.stride(0),
                                           None, None, L.stream(dev)))
    assert torch.equal(x_o[:, :O].cpu(), want_raw)


# --------------------------------------------------------------------------- n-step
@pytest.mark.parametrize("name", ["kat3", "n3", "n5", "n1"])
def test_nstep_golden(golden, dev, name):
    from pql_amd.replay.nstep_replay import NStepReplay
    g = golden("nstep")
    meta = g[f"nstep_{name}_meta"]
    N, n, O, A = (int(v) for v in meta[:4])
    ns = NStepReplay((O,), A, N, n, device=dev)
    for ci, Tn in enumerate(int(v) for v in meta[4:]):
        seed = 500 + 10 * ci
        obs = dd.uniform((N, Tn, O), seed); act = dd.uniform((N, Tn, A), seed + 1)
        rew = dd.uniform((N, Tn, 1), seed + 2); nobs = dd.uniform((N, Tn, O), seed + 3)
        done = dd.bernoulli((N, Tn, 1), seed + 4, 0.25)
        if name == "kat3":
            rew, done = g["nstep_kat3_in_rew"], g["nstep_kat3_in_done"]
        res = ns.add_to_buffer(*(T(a).to(dev) for a in (obs, act, rew, nobs, done)))
        for nm, t in zip(("obs", "act", "rew", "nobs", "done"), res):
            exp = g[f"nstep_{name}_c{ci}_{nm}"]
            assert tuple(t.shape) == exp.shape, (nm, ci)
            assert np.array_equal(t.cpu().numpy(), exp), (nm, ci)   # bit-exact incl. the fp32 reward sum


def test_nstep_first_call_too_short(dev):
    from pql_amd.replay.nstep_replay import NStepReplay
    ns = NStepReplay((2,), 1, 3, 3, device=dev)
    z = lambda *s: torch.zeros(*s, device=dev)  # noqa: E731
    with pytest.raises(RuntimeError):
        ns.add_to_buffer(z(3, 2, 2), z(3, 2, 1), z(3, 2, 1), z(3, 2, 2), z(3, 2, 1))


def test_nstep_large_vs_oracle(dev, ref):
    """4096 envs, warm-up T=32 then T=1 calls: every emitted row equals the oracle's."""
    from pql_amd.replay.nstep_replay import NStepReplay
    N, n, O, A = 4096, 3, 88, 16
    ns = NStepReplay((O,), A, N, n, device=dev)
    orc = ref.NStepRef(O, A, N, n)
    for ci, Tn in enumerate([32, 1, 1]):
        s = 9000 + 10 * ci
        args = [T(dd.uniform((N, Tn, O), s)), T(dd.uniform((N, Tn, A), s + 1)), T(dd.uniform((N, Tn, 1), s + 2)),
                T(dd.uniform((N, Tn, O), s + 3)), T(dd.bernoulli((N, Tn, 1), s + 4, 0.05))]
        got = ns.add_to_buffer(*(a.to(dev) for a in args))
        exp = orc.add(*args)
        for a, b in zip(got, exp):
            assert torch.equal(a.cpu(), b)


@pytest.mark.parametrize("seed", range(6))
def test_nstep_random_call_sequences_vs_oracle(dev, ref, seed):
    """Seeded random (envs, n, widths, gamma) and call sequences -- a first call of at least n steps, then calls of 1 .. 2n steps,
    dense and sparse dones -- against the oracle's per-window restatement of nstep_replay.py:74-92: every emitted row bit for bit
    (incl. the fp32 order of the discounted reward sum)."""
    from pql_amd.replay.nstep_replay import NStepReplay
    rs = np.random.RandomState(2000 + seed)
    N, n = int(rs.choice([1, 5, 64, 257])), int(rs.choice([1, 2, 3, 5, 7]))
    O, A = int(rs.choice([1, 4, 13])), int(rs.choice([1, 3, 6]))
    gamma = float(rs.choice([0.99, 0.9, 1.0]))
    ns = NStepReplay((O,), A, N, n, gamma=gamma, device=dev)
    orc = ref.NStepRef(O, A, N, n, gamma=gamma)
    p_done = float(rs.choice([0.02, 0.3, 0.9]))
    for ci in range(6):
        Tn = int(rs.randint(n, 3 * n + 1)) if ci == 0 else int(rs.randint(1, 2 * n + 1))
        s = 7000 + 100 * seed + 10 * ci
        args = [T(dd.uniform((N, Tn, O), s)), T(dd.uniform((N, Tn, A), s + 1)), T(dd.uniform((N, Tn, 1), s + 2, -2, 2)),
                T(dd.uniform((N, Tn, O), s + 3)), T(dd.bernoulli((N, Tn, 1), s + 4, p_done))]
        got = ns.add_to_buffer(*(a.to(dev) for a in args))
        exp = orc.add(*args)
        for nm, a, b in zip(("obs", "act", "rew", "nobs", "done"), got, exp):
            assert a.shape == b.shape and torch.equal(a.cpu(), b), (ci, nm)


# --------------------------------------------------------------------------- MLP family vs golden
SHAPES = [(8, 2), (88, 16), (211, 20), (108, 21)]


def _sd(state):
    return {k: T(v) for k, v in state.items()}


def _check_grads(module, garena, g, prefix, rtol=2e-4, atol=2e-6):
    lay = module.layout
    for n, pre in enumerate(module.key_prefixes):
        for l in range(lay.n_layers):
            for kind, view in (("weight", lay.weight(garena, n, l)), ("bias", lay.bias(garena, n, l))):
                key = f"{pre}{2 * l}.{kind}"
                np.testing.assert_allclose(dd.summarize(view.cpu().numpy()), g[f"{prefix}{key}"], rtol=rtol, atol=atol,
                                           err_msg=key)


@pytest.mark.parametrize("O,A", SHAPES)
def test_actor_module(golden, dev, O, A):
    from pql_amd.models.mlp import TanhMLPPolicy
    g = golden("models"); tag = f"o{O}a{A}"; B = 33 if O == 8 else 17
    m = TanhMLPPolicy((O,), A).to(dev); m.load_state_dict(_sd(dd.mlp_state(O, A, 11)))
    obs = T(dd.uniform((B, O), 1000 + O, -2, 2)).to(dev).requires_grad_(True)
    y = m(obs)
    np.testing.assert_allclose(y.detach().cpu().numpy(), g[f"actor_{tag}_y"], atol=1e-5)
    (y * T(dd.uniform((B, A), 3000 + O)).to(dev)).sum().backward()
    np.testing.assert_allclose(obs.grad.cpu().numpy(), g[f"actor_{tag}_dobs"], atol=2e-6)
    _check_grads(m, m.arena.grad, g, f"actor_{tag}_g_")
    # padding invariant: pad columns of W and pad entries of b carry exactly zero gradient
    in_views = sum(v.abs().sum().item() for v in (m.layout.weight(m.arena.grad, 0, l) for l in range(m.layout.n_layers)))
    in_views += sum(m.layout.bias(m.arena.grad, 0, l).abs().sum().item() for l in range(m.layout.n_layers))
    np.testing.assert_allclose(m.arena.grad.abs().sum().item(), in_views, rtol=1e-5)


@pytest.mark.parametrize("O,A", SHAPES)
def test_doubleq_module(golden, dev, O, A):
    from pql_amd.models.mlp import DoubleQ
    g = golden("models"); tag = f"o{O}a{A}"; B = 33 if O == 8 else 17
    q = DoubleQ((O,), A).to(dev); q.load_state_dict(_sd(dd.doubleq_state(O, A, 1, 21)))
    obs = T(dd.uniform((B, O), 1000 + O, -2, 2)).to(dev).requires_grad_(True)
    act = T(dd.uniform((B, A), 2000 + O)).to(dev).requires_grad_(True)
    q1, q2 = q.get_q1_q2(obs, act)
    np.testing.assert_allclose(q1.detach().cpu().numpy(), g[f"dq_{tag}_q1"], atol=1e-5)   # north_star bar
    np.testing.assert_allclose(q2.detach().cpu().numpy(), g[f"dq_{tag}_q2"], atol=1e-5)
    tgt = T(dd.uniform((B, 1), 4000 + O)).to(dev)
    loss = torch.nn.functional.mse_loss(q1, tgt) + torch.nn.functional.mse_loss(q2, tgt)
    np.testing.assert_allclose(loss.item(), g[f"dq_{tag}_loss"], rtol=1e-5)
    loss.backward()
    np.testing.assert_allclose(obs.grad.cpu().numpy(), g[f"dq_{tag}_dobs"], atol=2e-6)
    np.testing.assert_allclose(act.grad.cpu().numpy(), g[f"dq_{tag}_dact"], atol=2e-6)
    _check_grads(q, q.arena.grad, g, f"dq_{tag}_g_")
    o2 = obs.detach().clone().requires_grad_(True); a2 = act.detach().clone().requires_grad_(True)
    (-q.get_q_min(o2, a2).mean()).backward()
    np.testing.assert_allclose(a2.grad.cpu().numpy(), g[f"dq_{tag}_dpg_dact"], atol=2e-7)
    np.testing.assert_allclose(o2.grad.cpu().numpy(), g[f"dq_{tag}_dpg_dobs"], atol=2e-7)
    sd = q.state_dict()
    assert list(sd)[:2] == ["net_q1.net.0.weight", "net_q1.net.0.bias"] and len(sd) == 16


@pytest.mark.parametrize("O,A", SHAPES)
def test_distributional_module(golden, dev, O, A):
    from pql_amd.models.mlp import DistributionalDoubleQ
    g = golden("models"); tag = f"o{O}a{A}"; B = 33 if O == 8 else 17; K = 51
    q = DistributionalDoubleQ((O,), A, v_min=-10, v_max=10, num_atoms=K, device=dev).to(dev)
    q.load_state_dict(_sd(dd.doubleq_state(O, A, K, 31)))
    assert np.array_equal(q.z_atoms.cpu().numpy(), g[f"ddq_{tag}_z"])
    obs = T(dd.uniform((B, O), 1000 + O, -2, 2)).to(dev).requires_grad_(True)
    act = T(dd.uniform((B, A), 2000 + O)).to(dev).requires_grad_(True)
    p1, p2 = q.get_q1_q2(obs, act)
    np.testing.assert_allclose(p1.detach().cpu().numpy(), g[f"ddq_{tag}_p1"], atol=1e-6)
    np.testing.assert_allclose(p2.detach().cpu().numpy(), g[f"ddq_{tag}_p2"], atol=1e-6)
    np.testing.assert_allclose(q.get_q_min(obs, act).detach().cpu().numpy(), g[f"ddq_{tag}_qmin"], atol=1e-5)
    tg = T(g[f"ddq_{tag}_tgt"]).to(dev)
    loss = torch.nn.functional.binary_cross_entropy(p1, tg) + torch.nn.functional.binary_cross_entropy(p2, tg)
    loss.backward()
    np.testing.assert_allclose(obs.grad.cpu().numpy(), g[f"ddq_{tag}_dobs"], atol=2e-6)
    _check_grads(q, q.arena.grad, g, f"ddq_{tag}_g_")


def test_baseline_hidden_shape(golden, dev):
    from pql_amd.models.mlp import MLPNet
    g = golden("models"); hid = (512, 512, 256)
    net = MLPNet(104, 1, hidden_layers=list(hid)).to(dev); net.load_state_dict(_sd(dd.mlp_state(104, 1, 41, hid)))
    x = T(dd.uniform((9, 104), 6000)).to(dev).requires_grad_(True)
    y = net(x); y.sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["mlp_h512_512_256_y"], atol=1e-5)
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["mlp_h512_512_256_dx"], atol=2e-6)


def test_reference_checkpoint_file_through_the_hip_mlp(golden, dev, tmp_path):
    """f1: a checkpoint FILE in the reference's format (pql/utils/model_util.py:24-36) holding the weights of the reference's
    own pql/model.pth (63 -> 512 -> 256 -> 128 -> 12 actor with its extra `logstd`, `critic.net.*` critic) goes through
    `load_model` into the HIP MLPNet: outputs within 1e-5 and gradients within 2e-4 relative of the reference MLPNet's."""
    from pql_amd.models.mlp import MLPNet
    from pql_amd.utils.model_util import load_model
    g = golden("ckpt")
    actor_sd = {k[len("actor_w_"):]: T(g[k]) for k in g if k.startswith("actor_w_")}
    actor_sd["logstd"] = T(g["actor_logstd"])                                       # unexpected key: ignored, like strict=False
    critic_sd = {k[len("critic_w_"):]: T(g[k]) for k in g if k.startswith("critic_w_")}
    path = str(tmp_path / "model.pth")
    torch.save({"obs_rms": None, "actor": actor_sd, "critic": critic_sd}, path)
    for role, out_dim in (("actor", 12), ("critic", 1)):
        net = MLPNet(63, out_dim).to(dev)
        assert load_model(net, role, path)
        x = T(dd.uniform((19, 63), 7000 + out_dim, -2, 2)).to(dev).requires_grad_(True)
        y = net(x)
        (y * T(dd.uniform((19, out_dim), 7100 + out_dim)).to(dev)).sum().backward()
        np.testing.assert_allclose(y.detach().cpu().numpy(), g[f"{role}_y"], atol=1e-5)
        # (trained weights: input gradients of magnitude ~1, so the bar is relative -- 1e-5 -- where the synthetic nets' is absolute)
        np.testing.assert_allclose(x.grad.cpu().numpy(), g[f"{role}_dx"], rtol=1e-5, atol=2e-6)
        _check_grads(net, net.arena.grad, g, f"{role}_g_")
    assert load_model(object(), "obs_rms", path) is False                          # the file carries no statistics


@pytest.mark.parametrize("B", [8192, 777])
def test_mlp_full_batch_vs_oracle(dev, ref, B):
    """BASELINE batch (8192) and a ragged batch: forward within 1e-5, gradients at 1e-4 relative of the
    oracle (torch CPU autograd) -- exercises the 128x128 tiles, split-batch dW and tail handling."""
    from pql_amd.models.mlp import DoubleQ
    O, A = 88, 16
    q = DoubleQ((O,), A).to(dev); st = dd.doubleq_state(O, A, 1, 21); q.load_state_dict(_sd(st))
    obs = T(dd.uniform((B, O), 71, -2, 2)); act = T(dd.uniform((B, A), 72))
    w = T(dd.uniform((2, B, 1), 73))
    od = obs.to(dev).requires_grad_(True); ad = act.to(dev).requires_grad_(True)
    y = q._heads(od, ad)
    (y * w.to(dev)).sum().backward()
    q1 = [p.requires_grad_(True) for p in ref.params_from_state(st, "net_q1.net.")]
    q2 = [p.requires_grad_(True) for p in ref.params_from_state(st, "net_q2.net.")]
    oc = obs.clone().requires_grad_(True); ac = act.clone().requires_grad_(True)
    a, b = ref.twin_forward_ref(q1, q2, oc, ac)
    np.testing.assert_allclose(y[0].detach().cpu().numpy(), a.detach().numpy(), atol=1e-5)
    np.testing.assert_allclose(y[1].detach().cpu().numpy(), b.detach().numpy(), atol=1e-5)
    gr = torch.autograd.grad((a * w[0]).sum() + (b * w[1]).sum(), [oc, ac, *q1, *q2])
    np.testing.assert_allclose(od.grad.cpu().numpy(), gr[0].numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ad.grad.cpu().numpy(), gr[1].numpy(), rtol=1e-4, atol=1e-5)
    lay = q.layout
    k = 2
    for n in range(2):
        for l in range(lay.n_layers):
            gw, gb = gr[k].numpy(), gr[k + 1].numpy(); k += 2
            scale = np.abs(gw).max() + 1e-12
            np.testing.assert_allclose(lay.weight(q.arena.grad, n, l).cpu().numpy(), gw, rtol=1e-4, atol=2e-5 * scale)
            np.testing.assert_allclose(lay.bias(q.arena.grad, n, l).cpu().numpy(), gb, rtol=1e-4, atol=2e-5 * scale)


# --------------------------------------------------------------------------- losses
def test_c51_projection_golden(golden, dev):
    from pql_amd.utils.distl_util import projection
    g = golden("math")
    z = torch.linspace(-10, 10, 51, device=dev)
    out = projection(T(g["proj_p"]).to(dev), T(g["proj_rew"]).to(dev), T(g["proj_done"]).to(dev), float(g["proj_gamma"]),
                     -10, 10, 51, z, device=dev)
    np.testing.assert_allclose(out.cpu().numpy(), g["proj_out"], atol=1e-7)
    assert np.array_equal(out.cpu().numpy() != 0, g["proj_out"] != 0)   # same bins incl. integral-b / clamp / done rows
    z2 = torch.linspace(-2, 6, 11, device=dev)
    out2 = projection(T(g["proj2_p"]).to(dev), T(g["proj2_rew"]).to(dev), T(g["proj2_done"]).to(dev), 0.95, -2, 6, 11, z2)
    np.testing.assert_allclose(out2.cpu().numpy(), g["proj2_out"], atol=1e-7)


def _pad(x, ld):
    out = torch.zeros((*x.shape[:-1], ld)); out[..., : x.shape[-1]] = x
    return out


@pytest.mark.parametrize("B", [64, 8192])
def test_td_mse_loss(dev, ref, B):
    from pql_amd import _lib as L
    q = T(dd.uniform((2, B, 1), 81, -3, 3)); qt = T(dd.uniform((2, B, 1), 82, -3, 3))
    rew = T(dd.uniform((B,), 83, -1, 1)); done = T(dd.bernoulli((B,), 84, 0.2)); gn = float(np.float32(0.99 ** 3))
    qr = q.clone().requires_grad_(True)
    y = rew.view(-1, 1) + (1 - done.view(-1, 1)) * gn * torch.min(qt[0], qt[1])
    loss = torch.nn.functional.mse_loss(qr[0], y) + torch.nn.functional.mse_loss(qr[1], y)
    loss.backward()
    ld = 32
    # scalar heads write column 0 only: the dy buffer is allocated zeroed once and its pads are never dirtied
    dy = torch.zeros((2, B, ld), device=dev); lo = torch.zeros(1, device=dev); scr = torch.zeros(2048, device=dev)
    qd, qtd, rd, dd_ = _pad(q, ld).to(dev), _pad(qt, ld).to(dev), rew.to(dev), done.to(dev)
    L.check(L.lib.pqlk_td_mse_loss(L.ptr(qd), L.ptr(qtd), ld, L.ptr(rd),
                                   L.ptr(dd_), gn, B, L.ptr(dy), L.ptr(lo), None, 0, L.ptr(scr), L.stream(dev)))
    np.testing.assert_allclose(lo.item(), loss.item(), rtol=2e-6)
    np.testing.assert_allclose(dy[:, :, :1].cpu().numpy(), qr.grad.numpy(), rtol=1e-6, atol=1e-10)
    assert torch.all(dy[:, :, 1:] == 0)


@pytest.mark.parametrize("B", [37, 4096])
def test_c51_bce_loss(dev, ref, B):
    from pql_amd import _lib as L
    K, ld = 51, 64
    lg = T(dd.uniform((2, B, K), 91, -3, 3)); lt = T(dd.uniform((2, B, K), 92, -3, 3))
    rew = T(dd.uniform((B, 1), 93, -2, 2)); done = T(dd.bernoulli((B, 1), 94, 0.2)); gn = float(np.float32(0.99 ** 3))
    z = torch.linspace(-10, 10, K)
    lr = lg.clone().requires_grad_(True)
    with torch.no_grad():
        tgt = torch.min(ref.c51_project_ref(torch.softmax(lt[0], 1), rew, done, gn, -10, 10, K),
                        ref.c51_project_ref(torch.softmax(lt[1], 1), rew, done, gn, -10, 10, K))
    F = torch.nn.functional
    loss = F.binary_cross_entropy(torch.softmax(lr[0], 1), tgt) + F.binary_cross_entropy(torch.softmax(lr[1], 1), tgt)
    loss.backward()
    dy = torch.full((2, B, ld), 5.0, device=dev); lo = torch.zeros(1, device=dev); scr = torch.zeros(2048, device=dev)
    pj = torch.empty((B, K), device=dev)
    lgd, ltd, rd, dd_, zd = _pad(lg, ld).to(dev), _pad(lt, ld).to(dev), rew.view(-1).to(dev), done.view(-1).to(dev), z.to(dev)
    L.check(L.lib.pqlk_c51_bce_loss(L.ptr(lgd), L.ptr(ltd), ld, K, L.ptr(rd),
                                    L.ptr(dd_), L.ptr(zd), gn, -10.0, 10.0, B, L.ptr(dy), L.ptr(lo), None, 0,
                                    L.ptr(pj), L.ptr(scr), L.stream(dev)))
    np.testing.assert_allclose(pj.cpu().numpy(), tgt.numpy(), atol=2e-7)
    np.testing.assert_allclose(lo.item(), loss.item(), rtol=5e-6)
    np.testing.assert_allclose(dy[:, :, :K].cpu().numpy(), lr.grad.numpy(), rtol=2e-4, atol=2e-9)
    assert torch.all(dy[:, :, K:] == 0)


@pytest.mark.parametrize("K", [1, 51])
def test_dpg_loss(dev, K):
    from pql_amd import _lib as L
    B = 300; ld = L.ld(K)
    q = T(dd.uniform((2, B, K), 95, -3, 3))
    q[1, :5] = q[0, :5]  # ties -> gradient split evenly (torch.min backward)
    z = torch.linspace(-10, 10, K) if K > 1 else None
    qr = q.clone().requires_grad_(True)
    if K == 1:
        loss = -torch.min(qr[0], qr[1]).mean()
    else:
        e = [(torch.softmax(qr[i], 1) * z).sum(1) for i in range(2)]
        loss = -torch.min(e[0], e[1]).mean()
    loss.backward()
    dy = (torch.zeros if K == 1 else lambda *a, **k: torch.full(*a, 5.0, **k))((2, B, ld), device=dev)
    lo = torch.zeros(1, device=dev); scr = torch.zeros(2048, device=dev)
    qd = _pad(q, ld).to(dev); zd = z.to(dev) if K > 1 else None
    L.check(L.lib.pqlk_dpg_loss(L.ptr(qd), ld, K, L.ptr(zd), B, L.ptr(dy), L.ptr(lo), None, 0,
                                L.ptr(scr), L.stream(dev)))
    np.testing.assert_allclose(lo.item(), loss.item(), rtol=5e-6)
    np.testing.assert_allclose(dy[:, :, :K].cpu().numpy(), qr.grad.numpy(), rtol=5e-5, atol=1e-9)
    assert torch.all(dy[:, :, K:] == 0)


# --------------------------------------------------------------------------- optimiser
def test_clip_adamw_polyak_trace(dev, ref):
    from pql_amd import _lib as L
    n = 70016
    p0 = T(dd.uniform((n,), 61, -0.1, 0.1)); t0 = T(dd.uniform((n,), 62, -0.1, 0.1))
    opt = ref.AdamWRef([p0.clone()], lr=5e-4)
    tgt = [t0.clone()]
    p = p0.to(dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev); tg = t0.to(dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev); gn = torch.zeros(1, device=dev); scr = torch.zeros(2048, device=dev)
    for s in range(4):
        g = T(dd.uniform((n,), 63 + s, -1, 1)) * (10.0 if s % 2 == 0 else 1e-3)   # clipped and un-clipped steps
        norm_ref = torch.linalg.vector_norm(g).item()
        opt.apply([g.clone()], 0.5)
        ref.polyak_ref(tgt, opt.params, 0.05)
        gd = g.to(dev)
        L.check(L.lib.pqlk_clip_adamw_polyak(L.ptr(p), L.ptr(gd), L.ptr(m), L.ptr(v), L.ptr(tg), n, 1.0, 0.5, 5e-4, 0.9, 0.999,
                                             1e-8, 1e-2, 0.05, L.ptr(step), L.ptr(gn), L.ptr(scr), L.stream(dev)))
        np.testing.assert_allclose(gn.item(), norm_ref, rtol=1e-5)
        np.testing.assert_allclose(p.cpu().numpy(), opt.params[0].numpy(), rtol=2e-6, atol=1e-8)
        np.testing.assert_allclose(m.cpu().numpy(), opt.m[0].numpy(), rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(v.cpu().numpy(), opt.v[0].numpy(), rtol=1e-5, atol=1e-12)
        np.testing.assert_allclose(tg.cpu().numpy(), tgt[0].numpy(), rtol=2e-6, atol=1e-8)
    assert step.item() == 4


def test_batch_moments(dev):
    from pql_amd.utils.torch_util import RunningMeanStd
    x = T(dd.uniform((4096, 88), 55, -3, 5))
    rms = RunningMeanStd(shape=(88,), device=dev)
    bm, bv = rms.batch_moments(x.to(dev))
    np.testing.assert_allclose(bm.cpu().numpy(), x.mean(0).numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(bv.cpu().numpy(), x.var(0).numpy(), rtol=1e-5)


# --------------------------------------------------------------------------- fused hidden-layer forward
@pytest.mark.parametrize("dims,nets,B", [([104, 512, 512, 256, 1], 2, 8192),    # 64-row blocks, XCD-aware placement
                                          ([104, 512, 512, 256, 1], 2, 8131),    # 64-row blocks, ragged last tile
                                          ([88, 512, 256, 128, 16], 1, 777),     # 32-row blocks, ragged, idle waves at 128
                                          ([231, 512, 256, 128, 51], 2, 100),
                                          ([10, 32, 64, 2], 1, 33),              # reduction shorter than the weight ring
                                          ([40, 1024, 768, 288, 3], 1, 500),     # four / three tiles per wave, odd tile count
                                          ([72, 96, 1], 2, 16384)])              # one hidden layer
def test_fused_forward_equals_per_layer_path(dev, dims, nets, B):
    """k_mlp_fwd_fused (activations resident in LDS, fragment-ordered weights) accumulates every element in the same
    order as the per-layer k_gemm path: outputs and every stashed activation must be bit-identical."""
    from pql_amd import _lib as L
    from pql_amd.models.mlp import ArenaLayout, PackedWeights, mlp_forward_raw, output_view
    lay = ArenaLayout(dims, nets)
    arena = torch.zeros(lay.total, device=dev)
    for n in range(nets):
        for l in range(lay.n_layers):
            bound = 1.0 / np.sqrt(dims[l])
            lay.weight(arena, n, l).copy_(T(dd.uniform((dims[l + 1], dims[l]), 100 * n + l, -bound, bound)))
            lay.bias(arena, n, l).copy_(T(dd.uniform((dims[l + 1],), 100 * n + l + 50, -bound, bound)))
    x = torch.zeros((B, lay.ld_in), device=dev)
    x[:, : dims[0]] = T(dd.uniform((B, dims[0]), 7, -2, 2)).to(dev)
    pk = PackedWeights(lay, dev)
    assert pk.tensor is not None and pk.tensor.numel() == nets * sum(dims[l + 1] * L.ld(dims[l]) for l in range(lay.n_layers - 1))
    pk.refresh(arena)
    a_ref = mlp_forward_raw(lay, arena, x, L.ACT_NONE)
    a_fused = mlp_forward_raw(lay, arena, x, L.ACT_NONE, packed=pk, stash_all=True)
    # hidden stashes: bit-identical.  Output layer: when it rides inside the fused launch (<= 32 outputs) its reduction is
    # split over the 8 waves, i.e. a reassociation of the same products -> 1e-6 relative to the row's magnitude
    o_ref, o_fused = output_view(lay, a_ref, B), output_view(lay, a_fused, B)
    n_hidden_floats = a_ref.numel() - o_ref.numel()
    assert torch.equal(a_ref[:n_hidden_floats], a_fused[:n_hidden_floats])
    torch.testing.assert_close(o_fused, o_ref, rtol=2e-6, atol=2e-6)
    assert torch.equal(o_fused[:, :, dims[-1]:], o_ref[:, :, dims[-1]:])          # pad columns are zero in both
    a_inf = mlp_forward_raw(lay, arena, x, L.ACT_NONE, packed=pk, stash_all=False)   # inference: only the tail is stashed
    assert torch.equal(output_view(lay, a_inf, B), o_fused)


def test_fused_path_is_declined_for_unsupported_widths(dev):
    from pql_amd.models.mlp import ArenaLayout, PackedWeights
    assert PackedWeights(ArenaLayout([8, 100, 64, 1], 1), dev).tensor is None      # hidden width not a multiple of 32
    assert PackedWeights(ArenaLayout([8, 1056, 64, 1], 1), dev).tensor is None     # more than four output tiles per wave
    assert PackedWeights(ArenaLayout([2000, 64, 64, 1], 1), dev).tensor is None    # a 32-row input tile > 160 KB of LDS
    assert PackedWeights(ArenaLayout([8, 1], 1), dev).tensor is None               # no hidden layer


def test_optimizer_refreshes_packed_weight_copies(dev):
    """pqlk_clip_adamw_polyak_pack must leave packed_p == pack(new params) and packed_t == pack(new target), and the
    same parameters as the plain optimiser entry point."""
    from pql_amd import _lib as L
    from pql_amd.models.mlp import ArenaLayout, PackedWeights
    lay = ArenaLayout([104, 512, 256, 128, 1], 2)
    n = lay.total
    p0 = T(dd.uniform((n,), 1, -0.1, 0.1)).to(dev); t0 = T(dd.uniform((n,), 2, -0.1, 0.1)).to(dev)
    g = T(dd.uniform((n,), 3, -1, 1)).to(dev)
    outs = []
    for packing in (False, True):
        p, tg = p0.clone(), t0.clone()
        m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
        step = torch.zeros(1, dtype=torch.int32, device=dev); scr = torch.zeros(2048, device=dev)
        pk_p, pk_t = PackedWeights(lay, dev), PackedWeights(lay, dev)
        if packing:
            L.check(L.lib.pqlk_clip_adamw_polyak_pack(C.byref(lay.desc), L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), L.ptr(tg),
                                                      L.ptr(pk_p.tensor), L.ptr(pk_t.tensor), 1.0, 0.5, 5e-4, 0.9, 0.999, 1e-8, 1e-2,
                                                      0.05, L.ptr(step), None, L.ptr(scr), L.stream(dev)))
            ref_p, ref_t = PackedWeights(lay, dev).refresh(p).tensor, PackedWeights(lay, dev).refresh(tg).tensor
            assert torch.equal(pk_p.tensor, ref_p) and torch.equal(pk_t.tensor, ref_t)
        else:
            L.check(L.lib.pqlk_clip_adamw_polyak(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), L.ptr(tg), n, 1.0, 0.5, 5e-4, 0.9, 0.999,
                                                 1e-8, 1e-2, 0.05, L.ptr(step), None, L.ptr(scr), L.stream(dev)))
        outs.append((p, tg, m, v))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_fused_forward_ignores_columns_past_the_input_width(dev):
    """The fused path masks x[:, dims[0]:] while staging, so the target actor can read norm(next_obs) out of the wider
    [obs | action] tile whose extra columns are NOT zero."""
    from pql_amd import _lib as L
    from pql_amd.models.mlp import ArenaLayout, PackedWeights, mlp_forward_raw, output_view
    dims, B = [88, 512, 256, 128, 16], 300
    lay = ArenaLayout(dims, 1)
    arena = torch.zeros(lay.total, device=dev)
    for l in range(lay.n_layers):
        bound = 1.0 / np.sqrt(dims[l])
        lay.weight(arena, 0, l).copy_(T(dd.uniform((dims[l + 1], dims[l]), l, -bound, bound)))
        lay.bias(arena, 0, l).copy_(T(dd.uniform((dims[l + 1],), l + 50, -bound, bound)))
    pk = PackedWeights(lay, dev).refresh(arena)
    obs = T(dd.uniform((B, 88), 7, -2, 2)).to(dev)
    clean = torch.zeros((B, 96), device=dev); clean[:, :88] = obs
    wide = torch.full((B, 128), 1e30, device=dev); wide[:, :88] = obs          # "action" columns hold garbage
    y0 = output_view(lay, mlp_forward_raw(lay, arena, clean, L.ACT_TANH, packed=pk, stash_all=False), B).clone()
    y1 = output_view(lay, mlp_forward_raw(lay, arena, wide, L.ACT_TANH, packed=pk, stash_all=False), B)
    assert torch.equal(y0, y1) and torch.isfinite(y1).all()


# --------------------------------------------------------------------------- synthetic env (Isaac-Gym stand-in)
@pytest.mark.parametrize("n,O,A,off", [(257, 88, 16, 0), (4096, 211, 20, 4096), (33, 8, 2, 7)])
def test_synthetic_env_kernel_equals_torch_definition(dev, n, O, A, off):
    """The one-launch HIP env step computes the counter-based transition defined by the torch-op form: hashes,
    uniforms and `done` bit-exact; Box-Muller normals within libm-vs-device transcendental rounding (1e-5)."""
    from pql_amd.envs.synthetic import SyntheticVecEnv
    a = SyntheticVecEnv(n, O, A, device=dev, seed=1234, episode_length=30, env_offset=off)
    b = SyntheticVecEnv(n, O, A, device=dev, seed=1234, episode_length=30, env_offset=off)
    a.reset(), b.reset()
    for t in range(3):
        act = torch.from_numpy(dd.uniform((n, A), 50 + t) * 2 - 1).to(dev)
        o1, r1, d1, i1 = a.step(act)           # HIP
        b.t += 1
        o2, r2, d2, i2 = b._step_torch(act)    # torch ops on the same device
        torch.cuda.synchronize()
        assert d1.dtype == torch.bool and torch.equal(d1, d2)
        assert not bool(i1["TimeLimit.truncated"].any())
        np.testing.assert_allclose(o1.cpu().numpy(), o2.cpu().numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(r1.cpu().numpy(), r2.cpu().numpy(), rtol=1e-5, atol=1e-5)
    # shards of the env axis reproduce slices of the global env
    g = SyntheticVecEnv(2 * n, O, A, device=dev, seed=9, env_offset=0)
    s = SyntheticVecEnv(n, O, A, device=dev, seed=9, env_offset=n)
    g.reset(), s.reset()
    act = torch.from_numpy(dd.uniform((2 * n, A), 77) * 2 - 1).to(dev)
    og, rg, dg, _ = g.step(act)
    os_, rs, ds, _ = s.step(act[n:])
    assert torch.equal(og[n:], os_) and torch.equal(rg[n:], rs) and torch.equal(dg[n:], ds)


# --------------------------------------------------------------------------- SAC policy head
@pytest.mark.parametrize("tag", ["kat_toy", "kat_allegro"])
def test_squashed_gaussian_head_golden(golden, dev, tag):
    """pqlk_sg_head_forward / _backward (+ the MLP under them) vs the reference's TanhDiagGaussianMLPPolicy with the
    rsample draw injected: actions and log-prob within 1e-5 relative, parameter gradients of
    mean(0.3 logp - <a, w>) within the gradient bar used for the other heads.  kat_allegro has log_std beyond the
    +-5 clamp and u deep in tanh saturation (|u| ~ 300)."""
    from pql_amd import _lib as L
    from pql_amd.models.mlp import TanhDiagGaussianMLPPolicy, default_splits, mlp_forward_raw, output_view, pad_cols
    g = golden("sac")
    O, A, B = (int(v) for v in g[f"{tag}_meta"])
    state = dd.mlp_state(O, 2 * A, 61)
    if A > 2:
        state["net.6.bias"] = state["net.6.bias"].copy()
        state["net.6.bias"][A:] = np.linspace(-6.5, 6.5, A).astype(np.float32)
    pol = TanhDiagGaussianMLPPolicy((O,), A).to(dev)
    pol.load_state_dict({k: T(v) for k, v in state.items()})
    x = T(dd.uniform((B, O), 62, -2, 2)).to(dev)
    eps = T(g[f"{tag}_eps"]).to(dev)
    act, _, logp = pol.get_actions_logprob(x, eps=eps)
    np.testing.assert_allclose(act.cpu().numpy(), g[f"{tag}_act"], atol=1e-6)
    np.testing.assert_allclose(logp.cpu().numpy(), g[f"{tag}_logp"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(pol.get_actions(x, sample=False).cpu().numpy(), g[f"{tag}_mean_act"], atol=1e-6)
    np.testing.assert_allclose(pol(x).cpu().numpy(), g[f"{tag}_mean_act"], atol=1e-6)   # forward(sample=False)
    # backward of mean(0.3 logp - <a, w>): d/da = -w / B, d/dlogp = 0.3 / B
    lay = pol.layout
    x_pad = pad_cols(x, lay.ld_in)
    acts = mlp_forward_raw(lay, pol.arena.data, x_pad, L.ACT_NONE)
    y = output_view(lay, acts, B)[0]
    w = T(dd.uniform((A,), 63, -1, 1)).to(dev)
    da = (-w / B).repeat(B, 1).contiguous()
    dy = torch.full((1, B, lay.ld_out), 7.0, device=dev)
    L.check(L.lib.pqlk_sg_head_backward(L.ptr(y), lay.ld_out, L.ptr(eps), L.ptr(act), A, L.ptr(da), A, None, 0.3 / B, B, A, L.ptr(dy),
                                        L.stream(dev)))
    assert torch.all(dy[0, :, 2 * A:] == 0)
    splits = default_splits(B)
    grads = torch.empty_like(pol.arena.data); ws = torch.empty(lay.bwd_ws_floats(B, splits), device=dev)
    L.check(L.lib.pqlk_mlp_backward(C.byref(lay.desc), L.ptr(pol.arena.data), L.ptr(x_pad), lay.ld_in, B, L.ptr(acts), L.ptr(dy),
                                    L.ptr(grads), splits, None, 0, 0, 0, None, 0, L.ptr(ws), ws.numel(), L.stream(dev)))
    for l in range(lay.n_layers):
        for kind, view in (("weight", lay.weight(grads, 0, l)), ("bias", lay.bias(grads, 0, l))):
            np.testing.assert_allclose(dd.summarize(view.cpu().numpy()), g[f"{tag}_g_net.{2 * l}.{kind}"], rtol=2e-4, atol=2e-6,
                                       err_msg=f"{kind} {l}")
    np.testing.assert_allclose(lay.bias(grads, 0, lay.n_layers - 1).cpu().numpy(), g[f"{tag}_g_last_b"], rtol=2e-4, atol=2e-6)


# --------------------------------------------------------------------------- narrow kernels (narrow.h)
@pytest.mark.parametrize("O,A,hidden,B", [(88, 16, [512, 512, 256], 8192), (211, 20, [512, 256, 128], 1000), (8, 2, [64, 32], 77),
                                          (108, 21, [256, 256], 333)])
def test_dx_slice_kernel_equals_full_input_gradient(dev, O, A, hidden, B):
    """The DPG slice kernel (split-reduction MFMA, action columns only, tanh' fused) vs the generic path: the full input
    gradient from the same call with dx_tanh_of = NULL, sliced and multiplied by (1 - a^2) in torch.  The two differ only
    in summation order (4 partial sums vs one chain)."""
    from pql_amd import _lib as L
    from pql_amd.models.mlp import ArenaLayout, mlp_forward_raw
    lay = ArenaLayout([O + A, *hidden, 1], 2)
    arena = torch.zeros(lay.total, device=dev)
    for n in range(2):
        for l in range(lay.n_layers):
            bound = 1.0 / np.sqrt(lay.dims[l])
            lay.weight(arena, n, l).copy_(T(dd.uniform((lay.dims[l + 1], lay.dims[l]), 100 * n + l, -bound, bound)))
            lay.bias(arena, n, l).copy_(T(dd.uniform((lay.dims[l + 1],), 100 * n + l + 50, -bound, bound)))
    x = torch.zeros((B, lay.ld_in), device=dev)
    x[:, : O + A] = T(dd.uniform((B, O + A), 7, -1, 1)).to(dev)
    acts = mlp_forward_raw(lay, arena, x, L.ACT_NONE)
    dy = torch.zeros((2, B, lay.ld_out), device=dev)
    dy[:, :, 0] = T(dd.uniform((2, B), 9, -1, 1)).to(dev)
    a = torch.tanh(T(dd.uniform((B, A), 11, -2, 2))).to(dev).contiguous()
    ws = torch.empty(lay.bwd_ws_floats(B, 1), device=dev)
    full = torch.zeros((B, lay.ld_in), device=dev)
    L.check(L.lib.pqlk_mlp_backward(C.byref(lay.desc), L.ptr(arena), L.ptr(x), lay.ld_in, B, L.ptr(acts), L.ptr(dy), None, 1, L.ptr(full),
                                    lay.ld_in, 0, 0, None, 0, L.ptr(ws), ws.numel(), L.stream(dev)))
    ld_a = L.ld(A)
    sl = torch.full((B, ld_a), 3.0, device=dev)
    L.check(L.lib.pqlk_mlp_backward(C.byref(lay.desc), L.ptr(arena), L.ptr(x), lay.ld_in, B, L.ptr(acts), L.ptr(dy), None, 1, L.ptr(sl),
                                    ld_a, O, A, L.ptr(a), A, L.ptr(ws), ws.numel(), L.stream(dev)))
    want = full[:, O:O + A] * (1 - a * a)
    scale = float(want.abs().max())
    np.testing.assert_allclose(sl[:, :A].cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=2e-6 * scale)
    assert torch.all(sl[:, A:] == 3.0)   # only the slice columns are written


# --------------------------------------------------------------------------- shape sweep of the MLP path
_SWEEP = [
    # dims (in, hidden..., out), nets, B, fused?
    ([104, 512, 512, 256, 1], 2, 1024, True),     # BASELINE critic: fused body + fused head, 128x128 tiles, interior prefetch
    ([88, 512, 256, 128, 16], 1, 1000, True),     # actor, ragged batch
    ([229, 512, 256, 128, 51], 2, 96, True),      # C51 head (two narrow tiles), tiny batch
    ([60, 96, 64, 8], 1, 257, True),              # narrow layers: idle waves, short reductions
    ([48, 100, 36, 12], 1, 130, False),           # widths not multiples of 32: per-layer path, edge tiles everywhere
    ([17, 64, 33], 2, 513, True),                 # fused body, 33 outputs: narrow kernel with a ragged second tile
    ([129, 1024, 512, 5], 1, 300, True),          # 1024-wide layer (four tiles per wave)
    ([40, 32, 1], 2, 2049, True),                 # one tiny hidden layer, scalar head
]


@pytest.mark.parametrize("dims,nets,B,fused", _SWEEP)
def test_mlp_shape_sweep_vs_oracle(dev, ref, dims, nets, B, fused):
    """Forward (every stashed activation) and backward (parameter + input gradients) of arbitrary MLP shapes against the
    oracle's torch-CPU autograd: covers the fused / per-layer split, skinny / narrow / MFMA heads, interior and edge GEMM
    tiles, ragged batches."""
    from pql_amd import _lib as L
    from pql_amd.models.mlp import ArenaLayout, PackedWeights, default_splits, mlp_forward_raw, output_view
    lay = ArenaLayout(dims, nets)
    arena = torch.zeros(lay.total, device=dev)
    params = []
    for n in range(nets):
        net = []
        for l in range(lay.n_layers):
            bound = 1.0 / np.sqrt(dims[l])
            w = T(dd.uniform((dims[l + 1], dims[l]), 300 * n + l, -bound, bound)); b = T(dd.uniform((dims[l + 1],), 300 * n + l + 60, -bound, bound))
            lay.weight(arena, n, l).copy_(w); lay.bias(arena, n, l).copy_(b)
            net += [w.clone().requires_grad_(True), b.clone().requires_grad_(True)]
        params.append(net)
    xc = T(dd.uniform((B, dims[0]), 9, -2, 2)).requires_grad_(True)
    x = torch.zeros((B, lay.ld_in), device=dev); x[:, : dims[0]] = xc.detach().to(dev)
    pk = PackedWeights(lay, dev)
    assert (pk.tensor is not None) == fused
    if fused:
        pk.refresh(arena)
    acts = mlp_forward_raw(lay, arena, x, L.ACT_NONE, packed=pk if fused else None, stash_all=True)
    y = output_view(lay, acts, B)
    outs = [ref.mlp_forward_ref(params[n], xc) for n in range(nets)]
    for n in range(nets):
        np.testing.assert_allclose(y[n, :, : dims[-1]].cpu().numpy(), outs[n].detach().numpy(), rtol=1e-5, atol=1e-5)
        assert torch.all(y[n, :, dims[-1]:] == 0)
    # backward of sum_n <y_n, w_n>
    wts = [T(dd.uniform((B, dims[-1]), 70 + n, -1, 1)) for n in range(nets)]
    dy = torch.zeros((nets, B, lay.ld_out), device=dev)
    for n in range(nets):
        dy[n, :, : dims[-1]] = wts[n].to(dev)
    splits = default_splits(B)
    grads = torch.empty_like(arena); dx = torch.empty((B, lay.ld_in), device=dev)
    ws = torch.empty(lay.bwd_ws_floats(B, splits), device=dev)
    L.check(L.lib.pqlk_mlp_backward(C.byref(lay.desc), L.ptr(arena), L.ptr(x), lay.ld_in, B, L.ptr(acts), L.ptr(dy), L.ptr(grads), splits,
                                    L.ptr(dx), lay.ld_in, 0, 0, None, 0, L.ptr(ws), ws.numel(), L.stream(dev)))
    loss = sum((outs[n] * wts[n]).sum() for n in range(nets))
    gr = torch.autograd.grad(loss, [xc] + [p for net in params for p in net])
    gx = gr[0].numpy()
    np.testing.assert_allclose(dx[:, : dims[0]].cpu().numpy(), gx, rtol=1e-4, atol=2e-5 * (np.abs(gx).max() + 1e-12))
    k = 1
    for n in range(nets):
        for l in range(lay.n_layers):
            gw, gb = gr[k].numpy(), gr[k + 1].numpy(); k += 2
            scale = np.abs(gw).max() + 1e-12
            np.testing.assert_allclose(lay.weight(grads, n, l).cpu().numpy(), gw, rtol=1e-4, atol=2e-5 * scale, err_msg=f"dW net {n} layer {l}")
            np.testing.assert_allclose(lay.bias(grads, n, l).cpu().numpy(), gb, rtol=1e-4, atol=2e-5 * max(scale, np.abs(gb).max()), err_msg=f"db net {n} layer {l}")
    # data-parallel buckets (pqlk_mlp_backward_layers): the same chain run in layer ranges, each ending with the slab sum of its
    # own layers, leaves the same gradient bits -- the learner's bucket list and one bucket per layer
    from pql_amd.utils.dp import layer_buckets
    for buckets in (layer_buckets(lay.n_layers), [(l, l) for l in range(lay.n_layers - 1, -1, -1)]):
        g2 = torch.zeros_like(arena)
        for hi, lo in buckets:
            L.check(L.lib.pqlk_mlp_backward_layers(C.byref(lay.desc), L.ptr(arena), L.ptr(x), lay.ld_in, B, L.ptr(acts), L.ptr(dy), None, None,
                                                   None, 0.0, None, L.ptr(g2), splits, L.ptr(ws), ws.numel(), hi, lo, L.stream(dev)))
        for n in range(nets):
            for l in range(lay.n_layers):
                assert torch.equal(lay.weight(g2, n, l), lay.weight(grads, n, l)) and torch.equal(lay.bias(g2, n, l), lay.bias(grads, n, l)), (buckets, n, l)


# --------------------------------------------------------------------------- one-launch rollout bookkeeping
@pytest.mark.parametrize("N,O,A,win,H", [(1500, 13, 5, 7, 3), (4096, 88, 16, 100, 1), (64, 8, 2, 150, 2), (2049, 12, 4, 5, 2)])
def test_rollout_step_kernel_vs_reference_semantics(dev, N, O, A, win, H):
    """pqlk_rollout_step against a plain restatement of pql_actor.py:104-114,129-135 + common.py:195-202: slab columns,
    done' = done * !truncated, return / length accumulators, and the two moving windows as `deque.extend` of the finished envs'
    values in env order -- including steps where nobody, everybody, or more envs than the window holds finish, env counts
    that are not a multiple of the 1024-wide scan chunk, and field widths that are not multiples of 4."""
    from collections import deque
    from pql_amd import _lib as L
    rng = np.random.default_rng(N + O)
    f = dict(dtype=torch.float32, device=dev)
    sl = [torch.full((N, H, w), -7.0, **f) for w in (O, A, 1, O, 1)]
    cur_ret, cur_len = torch.zeros(N, **f), torch.zeros(N, **f)
    win_ret, win_len = torch.zeros(win + 1, **f), torch.zeros(win + 1, **f)
    ptr_r, ptr_l = torch.zeros(1, dtype=torch.int64, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
    ref_ret, ref_len = np.zeros(N, np.float32), np.zeros(N, np.float32)
    dq_r, dq_l = deque([0.0] * win, maxlen=win), deque([0.0] * win, maxlen=win)
    want = [np.full((N, H, w), -7.0, np.float32) for w in (O, A, 1, O, 1)]
    for rep in range(4):
        for t in range(H):
            p_done = (0.0, 0.02, 1.0, 0.4)[(rep + t) % 4]
            obs, act, nobs = (rng.normal(size=(N, w)).astype(np.float32) for w in (O, A, O))
            rew = rng.normal(size=N).astype(np.float32)
            done = rng.random(N) < p_done
            trunc = (rng.random(N) < 0.3) if rep % 2 else None
            d_obs, d_act, d_nobs, d_rew = (T(x).to(dev) for x in (obs, act, nobs, rew))
            d_done = T(done).to(dev)
            d_trunc = T(trunc).to(dev) if trunc is not None else None
            L.check(L.lib.pqlk_rollout_step(N, O, A, H, t, L.ptr(d_obs), L.ptr(d_act), L.ptr(d_nobs), L.ptr(d_rew),
                                            C.c_void_p(d_done.data_ptr()), C.c_void_p(d_trunc.data_ptr()) if d_trunc is not None else None,
                                            *[L.ptr(x) for x in sl], L.ptr(cur_ret), L.ptr(cur_len), L.ptr(win_ret), L.ptr(win_len),
                                            L.ptr(ptr_r), L.ptr(ptr_l), win, L.stream(dev)))
            torch.cuda.synchronize()
            want[0][:, t], want[1][:, t], want[2][:, t, 0], want[3][:, t] = obs, act, rew, nobs
            want[4][:, t, 0] = (done & ~trunc if trunc is not None else done).astype(np.float32)
            ref_ret += rew; ref_len += 1
            dq_r.extend(ref_ret[done].tolist()); dq_l.extend(ref_len[done].tolist())
            ref_ret[done] = 0; ref_len[done] = 0
            assert np.array_equal(cur_ret.cpu().numpy(), ref_ret) and np.array_equal(cur_len.cpu().numpy(), ref_len)
            assert sorted(win_ret[:win].cpu().tolist()) == sorted(np.float32(x) for x in dq_r)
            assert sorted(win_len[:win].cpu().tolist()) == sorted(np.float32(x) for x in dq_l)
            assert int(ptr_r.item()) == int(ptr_l.item()) < win
        for got, exp in zip(sl, want):
            assert np.array_equal(got.cpu().numpy(), exp)


# --------------------------------------------------------------------------- DPG backward: min-net compaction vs the dense chain
@pytest.mark.parametrize("hidden,B", [((512, 512, 256), 1000), ((512, 256, 128), 4096), ((128, 128), 130), ((512, 512, 256), 8192)])
def test_dpg_critic_backward_minnet_matches_dense_chain(dev, hidden, B):
    """pqlk_dpg_critic_backward partitions the batch by the net that attained min(Q1, Q2) and runs the dX chain over compact
    rows; pqlk_mlp_backward runs it densely over both nets with half the rows zero.  Same per-sample arithmetic, so the
    action gradients must agree to reassociation error of the final 512-term reduction -- including exact ties (both nets
    get half the gradient), a batch that is not a multiple of the 128-row tile, and every sample on one side."""
    from pql_amd import _lib as L
    from pql_amd.models.mlp import ArenaLayout, PackedWeights, mlp_forward_raw, output_view
    O, A = 88, 16
    lay = ArenaLayout([O + A, *hidden, 1], 2)
    g = torch.Generator(device=dev).manual_seed(B)
    arena = torch.zeros(lay.total, device=dev)
    for n in range(2):
        for l in range(lay.n_layers):
            bound = 1.0 / np.sqrt(lay.dims[l])
            lay.weight(arena, n, l).copy_((torch.rand(lay.weight(arena, n, l).shape, device=dev, generator=g) * 2 - 1) * bound)
            lay.bias(arena, n, l).copy_((torch.rand(lay.dims[l + 1], device=dev, generator=g) * 2 - 1) * bound)
    x = torch.zeros((B, lay.ld_in), device=dev)
    x[:, : O + A] = torch.randn((B, O + A), device=dev, generator=g)
    a_out = torch.zeros((1, B, L.ld(A)), device=dev)
    a_out[0, :, :A] = torch.tanh(torch.randn((B, A), device=dev, generator=g))
    pk = PackedWeights(lay, dev).refresh(arena)
    for case in ("natural", "ties", "all_net1"):
        acts = mlp_forward_raw(lay, arena, x, L.ACT_NONE, packed=pk, stash_all=True)
        q = output_view(lay, acts, B)
        if case == "ties":
            q[1, ::7, 0] = q[0, ::7, 0]          # exact ties on every 7th sample
        elif case == "all_net1":
            q[1, :, 0] = q[0, :, 0] - 1.0        # net 1 owns every sample: run 0 is empty
        dy = torch.zeros((2, B, lay.ld_out), device=dev)
        ring = torch.zeros(5, device=dev); slot = torch.zeros(1, dtype=torch.int32, device=dev); scratch = torch.zeros(2048, device=dev)
        owner = torch.zeros(B, dtype=torch.uint8, device=dev)
        L.check(L.lib.pqlk_dpg_loss_owner(L.ptr(q), lay.ld_out, 1, None, B, L.ptr(dy), L.ptr(ring), L.ptr(slot), 5, L.ptr(scratch),
                                          C.c_void_p(owner.data_ptr()), L.stream(dev)))
        outs = []
        for compact in (False, True, "no_owner"):
            dz = torch.full((1, B, L.ld(A)), 3.0 if compact else 0.0, device=dev)   # the compact path must zero it itself
            if compact:
                ws = torch.empty(int(L.lib.pqlk_dpg_backward_ws_floats(C.byref(lay.desc), B)), device=dev)
                L.check(L.lib.pqlk_dpg_critic_backward(C.byref(lay.desc), L.ptr(arena), L.ptr(x), lay.ld_in, B, L.ptr(acts), L.ptr(dy),
                                                       L.ptr(dz), L.ld(A), O, A, L.ptr(a_out), L.ld(A),
                                                       C.c_void_p(owner.data_ptr()) if compact is True else None,   # NULL: derived from q
                                                       L.ptr(ws), ws.numel(), L.stream(dev)))
            else:
                ws = torch.empty(lay.bwd_ws_floats(B, 1), device=dev)
                L.check(L.lib.pqlk_mlp_backward(C.byref(lay.desc), L.ptr(arena), L.ptr(x), lay.ld_in, B, L.ptr(acts), L.ptr(dy), None, 1,
                                                L.ptr(dz), L.ld(A), O, A, L.ptr(a_out), L.ld(A), L.ptr(ws), ws.numel(), L.stream(dev)))
            torch.cuda.synchronize()
            outs.append(dz.clone())
        dense, comp, comp_q = outs
        assert float(dense.abs().max()) > 0
        torch.testing.assert_close(comp[0, :, :A], dense[0, :, :A], rtol=2e-5, atol=1e-9 + 2e-6 * float(dense.abs().max()))
        assert torch.all(comp[0, :, A:] == 0)
        assert torch.equal(comp, comp_q)   # ownership from the byte array == ownership derived from the Q heads


@pytest.mark.parametrize("hidden,A,B", [([512, 512, 256], 16, 8192), ([512, 256, 128], 16, 1000), ([512, 256, 128], 2, 256), ([256, 128, 128], 5, 333)])
def test_dpg_backward_fused_matches_the_separate_launches(dev, hidden, A, B):
    """Round 4: pqlk_mlp_forward_qc + pqlk_dpg_backward_fused + pqlk_mlp_backward_tail (DPG loss, partition and compact head in one
    launch off the compact Q; the actor's head backward inside the action-slice launch) against the launches they replace
    (pqlk_dpg_loss_owner + pqlk_dpg_critic_backward + pqlk_mlp_backward): same loss, same action gradient, same actor gradient up to
    the reassociation of sums -- on natural Q values, with exact ties on every 7th sample (both nets own the sample: finished in
    the run-1 tile) and with every sample owned by net 1 (run 0 empty)."""
    from pql_amd import _lib as L
    from pql_amd.models.mlp import ArenaLayout, PackedWeights, default_splits, mlp_forward_raw, output_view
    O = 88
    cl, al = ArenaLayout([O + A, *hidden, 1], 2), ArenaLayout([O, *hidden, A], 1)
    assert L.lib.pqlk_dpg_fused_ok(C.byref(cl.desc), C.byref(al.desc), B) == 1
    g = torch.Generator(device=dev).manual_seed(B)
    carena, aarena = torch.zeros(cl.total, device=dev), torch.zeros(al.total, device=dev)
    for lay, arena in ((cl, carena), (al, aarena)):
        for n in range(lay.n_nets):
            for l in range(lay.n_layers):
                bound = 1.0 / np.sqrt(lay.dims[l])
                lay.weight(arena, n, l).copy_((torch.rand(lay.weight(arena, n, l).shape, device=dev, generator=g) * 2 - 1) * bound)
                lay.bias(arena, n, l).copy_((torch.rand(lay.dims[l + 1], device=dev, generator=g) * 2 - 1) * bound)
    x_obs = torch.zeros((B, al.ld_in), device=dev); x_obs[:, :O] = torch.randn((B, O), device=dev, generator=g)
    x_sa = torch.zeros((B, cl.ld_in), device=dev); x_sa[:, :O] = x_obs[:, :O]
    pkc, pka = PackedWeights(cl, dev).refresh(carena), PackedWeights(al, dev).refresh(aarena)
    acts_a = mlp_forward_raw(al, aarena, x_obs, L.ACT_TANH, out2=x_sa[:, O:], packed=pka, stash_all=True)
    a_out = output_view(al, acts_a, B)
    ld_a, splits, st = L.ld(A), default_splits(B), L.stream(dev)
    acts_c = torch.empty(cl.acts_floats(B), device=dev)
    qc = torch.zeros((2, B), device=dev)
    L.check(L.lib.pqlk_mlp_forward_qc(C.byref(cl.desc), L.ptr(carena), L.ptr(pkc.tensor), 1, L.ptr(x_sa), cl.ld_in, None, 0, 0, B, L.ptr(acts_c),
                                      L.ptr(qc), st))
    acts_ref = mlp_forward_raw(cl, carena, x_sa, L.ACT_NONE, packed=pkc, stash_all=True)
    assert torch.equal(acts_c, acts_ref)                                    # the compact copy changes nothing else
    assert torch.equal(qc, output_view(cl, acts_c, B)[:, :, 0])
    # split input: [obs | action] read from the actor's input tile and the actor's output block (no concatenated tile): same bits
    acts_s, qc_s = torch.empty_like(acts_c), torch.zeros_like(qc)
    L.check(L.lib.pqlk_mlp_forward_qc(C.byref(cl.desc), L.ptr(carena), L.ptr(pkc.tensor), 1, L.ptr(x_obs), al.ld_in, L.ptr(a_out), L.ld(A), O, B,
                                      L.ptr(acts_s), L.ptr(qc_s), st))
    assert torch.equal(acts_s, acts_c) and torch.equal(qc_s, qc)
    rc = L.lib.pqlk_mlp_forward_qc(C.byref(cl.desc), L.ptr(carena), L.ptr(pkc.tensor), 1, L.ptr(x_obs), al.ld_in, L.ptr(a_out), L.ld(A), O + 2, B,
                                   L.ptr(acts_s), L.ptr(qc_s), st)
    assert rc == 3   # PQLK_E_RANGE: the split must fall on a 16-byte boundary
    for case in ("natural", "ties", "all_net1"):
        q = output_view(cl, acts_c, B)
        if case == "ties":
            q[1, ::7, 0] = q[0, ::7, 0]
        elif case == "all_net1":
            q[1, :, 0] = q[0, :, 0] - 1.0
        qc.copy_(q[:, :, 0])
        # ---- the separate launches
        dy = torch.zeros((2, B, cl.ld_out), device=dev); owner = torch.zeros(B, dtype=torch.uint8, device=dev)
        ring = torch.zeros(5, device=dev); slot = torch.zeros(1, dtype=torch.int32, device=dev); scratch = torch.zeros(2048, device=dev)
        L.check(L.lib.pqlk_dpg_loss_owner(L.ptr(q), cl.ld_out, 1, None, B, L.ptr(dy), L.ptr(ring), L.ptr(slot), 5, L.ptr(scratch),
                                          C.c_void_p(owner.data_ptr()), st))
        dz_ref = torch.zeros((1, B, ld_a), device=dev)
        ws_c = torch.empty(int(L.lib.pqlk_dpg_backward_ws_floats(C.byref(cl.desc), B)), device=dev)
        L.check(L.lib.pqlk_dpg_critic_backward(C.byref(cl.desc), L.ptr(carena), L.ptr(x_sa), cl.ld_in, B, L.ptr(acts_c), L.ptr(dy), L.ptr(dz_ref),
                                               ld_a, O, A, L.ptr(a_out), ld_a, C.c_void_p(owner.data_ptr()), L.ptr(ws_c), ws_c.numel(), st))
        g_ref = torch.zeros(al.total, device=dev)
        ws_a = torch.empty(al.bwd_ws_floats(B, splits), device=dev)
        L.check(L.lib.pqlk_mlp_backward(C.byref(al.desc), L.ptr(aarena), L.ptr(x_obs), al.ld_in, B, L.ptr(acts_a), L.ptr(dz_ref), L.ptr(g_ref),
                                        splits, None, 0, 0, 0, None, 0, L.ptr(ws_a), ws_a.numel(), st))
        # ---- the fused launches (fresh workspaces, poisoned: nothing may depend on what they held)
        dz = torch.zeros((1, B, ld_a), device=dev); dz[0, :, :A] = 3.0
        ws_c2 = torch.full_like(ws_c, float("nan")); ws_a2 = torch.full_like(ws_a, float("nan"))
        parts = torch.full((64,), float("nan"), device=dev)
        L.check(L.lib.pqlk_dpg_backward_fused(C.byref(cl.desc), L.ptr(carena), L.ptr(x_sa), cl.ld_in, B, L.ptr(acts_c), L.ptr(qc), L.ptr(dz), ld_a, O,
                                              L.ptr(a_out), ld_a, L.ptr(parts), L.ptr(ws_c2), ws_c2.numel(), C.byref(al.desc), L.ptr(aarena),
                                              L.ptr(acts_a), L.ptr(ws_a2), ws_a2.numel(), splits, st))
        g_new = torch.full((al.total,), float("nan"), device=dev)
        mn_ptr = C.c_void_p(ws_c2.data_ptr() + 4 * int(L.lib.pqlk_dpg_fused_mn_offset(C.byref(cl.desc), B)))
        L.check(L.lib.pqlk_mlp_backward_tail(C.byref(al.desc), L.ptr(aarena), L.ptr(x_obs), al.ld_in, B, L.ptr(acts_a), L.ptr(g_new), splits,
                                             L.ptr(ws_a2), ws_a2.numel(), None, None, int(L.lib.pqlk_dpg_fused_head_parts(B)), mn_ptr, st))
        torch.cuda.synchronize()
        n_parts = int(L.lib.pqlk_dpg_fused_loss_parts())
        loss_new = float(parts[:n_parts].double().sum()) * (-1.0 / B)
        np.testing.assert_allclose(loss_new, float(ring[0]), rtol=2e-6, atol=1e-7, err_msg=case)
        scale = float(dz_ref.abs().max())
        assert scale > 0
        torch.testing.assert_close(dz[0, :, :A], dz_ref[0, :, :A], rtol=2e-5, atol=1e-9 + 2e-6 * scale, msg=case)
        assert torch.all(dz[0, :, A:] == 0)
        assert torch.isfinite(g_new).all()
        torch.testing.assert_close(g_new, g_ref, rtol=2e-5, atol=1e-9 + 2e-6 * float(g_ref.abs().max()), msg=case)
        mn = ws_c2.view(torch.int32)[int(L.lib.pqlk_dpg_fused_mn_offset(C.byref(cl.desc), B)):][:4].tolist()
        own = owner.cpu().numpy()
        assert mn[0] == int((own & 1).sum()) and mn[1] == int(((own >> 1) & 1).sum()) and mn[2] % 128 == 0 and mn[3] % 128 == 0


def test_output_only_forward_over_k_batches_equals_the_per_batch_forwards(dev):
    """algo.actor_ahead: the target policy's noised actions of the next K V-learner steps come from ONE forward launch over K x B rows
    (stash_all = PQLK_STASH_OUTPUT_ONLY: `acts` is the output block, nothing else is written).  At K x B = 65 536 rows the launch
    runs 64-row tiles (two row tiles share every weight fragment), the per-step launch of 8192 rows 32-row tiles: the same k-ordered
    fp32 fma chain per output element, so the actions -- in the output block and dropped into the critic-input tiles -- must be
    BIT-identical to K separate calls."""
    from pql_amd import _lib as L
    from pql_amd.models.mlp import ArenaLayout, PackedWeights, mlp_forward_raw, output_view
    O, A, B, K = 88, 16, 8192, 8
    lay = ArenaLayout([O, 512, 512, 256, A], 1)
    g = torch.Generator(device=dev).manual_seed(5)
    arena = torch.zeros(lay.total, device=dev)
    for l in range(lay.n_layers):
        bound = 1.0 / np.sqrt(lay.dims[l])
        lay.weight(arena, 0, l).copy_((torch.rand(lay.weight(arena, 0, l).shape, device=dev, generator=g) * 2 - 1) * bound)
        lay.bias(arena, 0, l).copy_((torch.rand(lay.dims[l + 1], device=dev, generator=g) * 2 - 1) * bound)
    pk = PackedWeights(lay, dev).refresh(arena)
    ld_sa = L.ld(O + A)
    tiles = torch.zeros((K, B, ld_sa), device=dev)
    tiles[:, :, :O] = torch.randn((K, B, O), device=dev, generator=g)
    draw = torch.randn((K, B, A), device=dev, generator=g)
    ref_tiles = tiles.clone()
    ref_out = []
    for k in range(K):
        acts = mlp_forward_raw(lay, arena, ref_tiles[k], L.ACT_TANH_NOISE, draw[k], 0.8, 0.2, out2=ref_tiles[k][:, O:], packed=pk, stash_all=False)
        ref_out.append(output_view(lay, acts, B)[0].clone())
    out = torch.full((K * B, L.ld(A)), 7.0, device=dev)
    flat = tiles.view(K * B, ld_sa)
    mlp_forward_raw(lay, arena, flat, L.ACT_TANH_NOISE, draw.view(K * B, A), 0.8, 0.2, out, flat[:, O:], packed=pk, stash_all=2)
    torch.cuda.synchronize()
    assert torch.equal(tiles, ref_tiles)
    assert torch.equal(out.view(K, B, -1), torch.stack(ref_out))
    assert float(tiles[:, :, O:O + A].abs().max()) <= 1.0 and float(tiles[:, :, O:O + A].abs().mean()) > 0.05
    # the mode needs the fused stack + fused head: refused otherwise instead of writing past a small buffer
    rc = L.lib.pqlk_mlp_forward(C.byref(lay.desc), L.ptr(arena), None, 2, L.ptr(flat), ld_sa, K * B, L.ACT_TANH, None, 0.0, 0.0, L.ptr(out), None, 0,
                                L.stream(dev))
    assert rc == 5   # PQLK_E_UNSUPPORTED


_GEMM_LOOP_SCRIPT = r"""
import ctypes as C, hashlib, sys
sys.path.insert(0, {tests!r}); sys.path.insert(0, {root!r})
import numpy as np, torch
import detdata as dd
from pql_amd import _lib as L
from pql_amd.models.mlp import ArenaLayout, PackedWeights, default_splits, mlp_forward_raw
dev = torch.device("cuda:0")
out = []
for dims, nets, B in (([104, 512, 512, 256, 1], 2, 8192), ([88, 512, 256, 128, 16], 1, 4096), ([232, 512, 512, 256, 51], 2, 2048)):
    lay = ArenaLayout(dims, nets)
    arena = torch.zeros(lay.total, device=dev)
    for n in range(nets):
        for l in range(lay.n_layers):
            lay.weight(arena, n, l).copy_(torch.from_numpy(dd.uniform((dims[l + 1], dims[l]), 40 * n + l, -0.05, 0.05)))
            lay.bias(arena, n, l).copy_(torch.from_numpy(dd.uniform((dims[l + 1],), 40 * n + l + 20, -0.05, 0.05)))
    x = torch.zeros((B, lay.ld_in), device=dev); x[:, :dims[0]] = torch.from_numpy(dd.uniform((B, dims[0]), 4243, -1, 1)).to(dev)
    pk = PackedWeights(lay, dev); pk.refresh(arena)
    acts = mlp_forward_raw(lay, arena, x, L.ACT_NONE, packed=pk, stash_all=True)
    dy = torch.zeros((nets, B, lay.ld_out), device=dev)
    dy[:, :, :dims[-1]] = torch.from_numpy(dd.uniform((nets, B, dims[-1]), 4244, -1e-3, 1e-3)).to(dev)
    splits = default_splits(B)
    grads = torch.empty_like(arena); dx = torch.empty((B, lay.ld_in), device=dev)
    ws = torch.empty(lay.bwd_ws_floats(B, splits), device=dev)
    L.check(L.lib.pqlk_mlp_backward(C.byref(lay.desc), L.ptr(arena), L.ptr(x), lay.ld_in, B, L.ptr(acts), L.ptr(dy), L.ptr(grads), splits,
                                    L.ptr(dx), lay.ld_in, 0, 0, None, 0, L.ptr(ws), ws.numel(), L.stream(dev)))
    torch.cuda.synchronize()
    assert torch.isfinite(grads).all() and grads.abs().max() > 0
    out.append(hashlib.sha256(grads.cpu().numpy().tobytes() + dx.cpu().numpy().tobytes()).hexdigest())
print(" ".join(out))
"""


def test_gemm_lds_dma_loop_is_bitwise_the_register_staged_loop(dev):
    """The backward GEMMs take the LDS-DMA main loop when every tile of the grid is interior and the register-staged one
    otherwise (gemm.hip, launch_gemm).  Same reduction order by construction; here: the gradients and the input gradient of
    three BASELINE-shaped MLPs hash identically with the DMA loop on and with PQLK_GEMM_DMA=0 / PQLK_GEMM_XCD=0 (the
    switches are read once per process, hence the subprocesses)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = _GEMM_LOOP_SCRIPT.format(tests=os.path.join(root, "tests"), root=root)
    hashes = {}
    for name, extra in (("dma+xcd", {}), ("staged", {"PQLK_GEMM_DMA": "0"}), ("dma, dispatch order", {"PQLK_GEMM_XCD": "0"})):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        hashes[name] = r.stdout.strip().split()
    assert len(hashes["dma+xcd"]) == 3
    assert hashes["dma+xcd"] == hashes["staged"] == hashes["dma, dispatch order"], hashes


# --------------------------------------------------------------------------- the rollout's elementwise launches
def test_rms_merge_normalize_and_action_noise_kernels_match_the_cpu_expressions(dev):
    """pqlk_rms_merge / pqlk_rms_normalize / pqlk_action_noise against the reference's torch expressions evaluated on the CPU
    (pql/utils/torch_util.py:83-103, noise.py:19-41): bit for bit -- same fp32 operations in the same order, IEEE division."""
    from pql_amd.utils.noise import add_mixed_normal_noise, add_normal_noise
    from pql_amd.utils.torch_util import RunningMeanStd
    O, N, A = 88, 4096, 16
    rms = RunningMeanStd(shape=(O,), device=dev)
    mean_c, var_c, count = torch.zeros(O), torch.ones(O), 1e-4
    for step in range(5):
        bm, bv = T(dd.uniform((O,), 600 + step, -1, 1)), T(dd.uniform((O,), 610 + step, 0.1, 3.0))
        n = 4096 if step else 131072      # warm-up batch of 32 steps, then one step
        rms.update_from_moments(bm.to(dev), bv.to(dev), n)
        delta = bm - mean_c               # the reference expression on the CPU
        tot = count + n
        m2 = var_c * count + bv * n + delta ** 2 * count * n / tot
        mean_c = mean_c + delta * n / tot
        var_c = m2 / tot
        count = tot
        assert torch.equal(rms.mean.cpu(), mean_c) and torch.equal(rms.var.cpu(), var_c) and rms.count == count
    x = T(dd.uniform((N, O), 620, -4, 4))
    # (sqrt through numpy: torch's vectorised CPU sqrt is 1 ulp off on ~0.6 % of inputs on some hosts)
    want = (x - mean_c) / torch.from_numpy(np.sqrt((var_c + 1e-4).numpy()))
    assert torch.equal(rms.normalize(x.to(dev)).cpu(), want)
    act, draw = T(dd.uniform((N, A), 630, -1, 1)), T(dd.uniform((N, A), 631, -3, 3))
    std = torch.linspace(0.05, 0.8, N).unsqueeze(-1)
    got = add_mixed_normal_noise(act.to(dev), std_max=0.8, std_min=0.05, out_bounds=[-1., 1.], draw=draw.to(dev))
    assert torch.equal(got.cpu(), (act + draw * std).clamp(-1., 1.))
    got = add_normal_noise(act.to(dev), std=0.3, out_bounds=[-1., 1.], draw=draw.to(dev))
    assert torch.equal(got.cpu(), (act + draw * 0.3).clamp(-1., 1.))
    # un-injected: the generator is consumed exactly like torch.normal(zeros, std) consumes it
    g1, g2 = torch.Generator(device=dev), torch.Generator(device=dev)
    g1.manual_seed(5); g2.manual_seed(5)
    a = add_mixed_normal_noise(act.to(dev), std_max=0.8, std_min=0.05, out_bounds=[-1., 1.], generator=g1)
    b = (act.to(dev) + torch.empty((N, A), device=dev).normal_(generator=g2) * std.to(dev)).clamp(-1., 1.)
    assert torch.equal(a, b)


# --------------------------------------------------------------------------- the learners' draws
def test_philox_draws_are_torchs_own_numbers(dev):
    """pqlk_philox_draws == the stream of torch calls it replaces, bit for bit: `randint(range, (B,))` then `(B, A).normal_()`
    per step on one device generator, K steps per launch, at BASELINE shapes (cfg #2: one value per Philox block; cfg #5: the
    normal tensor is wider than the 2048 x 256-thread launch, so the blocks' second components are used too), from a
    non-zero offset; and the offset a generator is left at."""
    from pql_amd.utils import rng as R
    contract = R.verified(dev)
    assert contract is not None, "pqlk_philox_draws does not reproduce torch.randint / normal_ on this device"
    for B, A, cap, K, seed, off0 in ((8192, 16, 1_000_000, 8, 20240229, 0), (32768, 21, 5_000_000, 3, 5, 64), (256, 2, 777, 5, 11, 4)):
        g = torch.Generator(device=dev); g.manual_seed(seed); g.set_offset(off0)
        wi, wn = [], []
        for _ in range(K):
            wi.append(torch.randint(cap, (B,), generator=g, device=dev))
            wn.append(torch.empty((B, A), device=dev).normal_(generator=g))
        g2 = torch.Generator(device=dev); g2.manual_seed(seed); g2.set_offset(off0)
        ahead = R.DrawAhead(g2, dev, B, (B, A), K, contract)
        ahead.refill(cap)
        assert torch.equal(ahead.idx, torch.stack(wi)) and torch.equal(ahead.normal, torch.stack(wn))
        assert int(ahead.idx.max()) < cap and int(ahead.idx.min()) >= 0
        for k in range(K):
            assert ahead.take() == k
        assert g2.get_offset() == g.get_offset()          # the generator object ends where K x 2 torch calls leave it
        # indices only (the P-learner's stream): K randint calls
        g3 = torch.Generator(device=dev); g3.manual_seed(seed + 1)
        wi = [torch.randint(cap, (B,), generator=g3, device=dev) for _ in range(K)]
        g4 = torch.Generator(device=dev); g4.manual_seed(seed + 1)
        a2 = R.DrawAhead(g4, dev, B, None, K, contract)
        a2.refill(cap)
        assert torch.equal(a2.idx, torch.stack(wi))


def test_first_calls_from_two_threads_at_once(dev):
    """ctypes drops the GIL: with algo.async_learners the V and P learner threads can make a kernel's FIRST call in the same
    instant.  The per-device once-flag must not let the second caller launch (with > 64 KB of dynamic LDS) before the first has
    raised the kernel's limit -- round 3 marked the device done before calling hipFuncSetAttribute.  Run in a fresh process so
    that the calls really are the first ones."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import ctypes as C, threading, sys, torch
sys.path.insert(0, %r)
from pql_amd import _lib as L
from pql_amd.models.mlp import DoubleQ, PackedWeights, mlp_forward_raw, output_view
dev = torch.device("cuda:0")
torch.manual_seed(3)
q = DoubleQ((88,), 16, hidden_layers=[512, 512, 256]).to(dev)
lay = q.layout
pk = PackedWeights(lay, dev)
B = 4096
x = torch.zeros(B, L.ld(104), device=dev); x[:, :104].normal_()
torch.cuda.synchronize()
streams = [torch.cuda.Stream(dev) for _ in range(2)]
gate = threading.Barrier(2)
outs, errs = [None, None], []
def run(i):
    try:
        with torch.cuda.stream(streams[i]):
            gate.wait()
            outs[i] = mlp_forward_raw(lay, q.arena.data, x, packed=None).clone()   # per-layer k_gemm launches: the process's first
    except Exception as e:   # noqa: BLE001
        errs.append(repr(e))
ts = [threading.Thread(target=run, args=(i,)) for i in range(2)]
[t.start() for t in ts]; [t.join() for t in ts]
torch.cuda.synchronize()
assert not errs, errs
assert torch.equal(outs[0], outs[1])
pk.refresh(q.arena.data); torch.cuda.synchronize()
gate2 = threading.Barrier(2)
def run2(i):
    try:
        with torch.cuda.stream(streams[i]):
            gate2.wait()
            outs[i] = mlp_forward_raw(lay, q.arena.data, x, packed=pk).clone()   # k_mlp_fwd_fused: 132 KB of dynamic LDS
    except Exception as e:   # noqa: BLE001
        errs.append(repr(e))
ts = [threading.Thread(target=run2, args=(i,)) for i in range(2)]
[t.start() for t in ts]; [t.join() for t in ts]
torch.cuda.synchronize()
assert not errs, errs
assert torch.equal(outs[0], outs[1])
ref = mlp_forward_raw(lay, q.arena.data, x, packed=pk); torch.cuda.synchronize()
assert torch.equal(output_view(lay, ref, B), output_view(lay, outs[0], B))
print("ok")
''' % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
