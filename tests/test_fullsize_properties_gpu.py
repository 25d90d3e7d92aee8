"""Size-independent properties of the HIP path at BASELINE.json's FULL sizes, where the CPU oracle is too slow to be the
checker: replay rings of 1 M / 5 M rows (cfg #2 / #5: the 5 M x 1 KiB ring is 5.1 GB, so byte offsets pass 2^32), batches
of 8192 / 32768, 16384 envs with 211-wide observations (cfg #4).  The checkers are round trips, conservation laws,
linearity and idempotence, evaluated with plain torch ops on the same device.  Run with `pytest -m gpu`."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rows(idx, width, salt):
    """Deterministic row content from the GLOBAL row number: value(r, c) = frac-ish hash in [-1, 1)."""
    r = idx.to(torch.int64).unsqueeze(1)
    c = torch.arange(width, device=idx.device, dtype=torch.int64).unsqueeze(0)
    h = (r * 2654435761 + c * 40503 + salt * 7919) & 0xFFFFFF
    return (h.to(torch.float32) / float(1 << 23)) - 1.0


@pytest.mark.parametrize("cap,O,A,B", [(1_000_000, 88, 16, 8192), (5_000_000, 108, 21, 32768)])
def test_replay_round_trip_full_ring(dev, cap, O, A, B):
    """insert -> gather is the identity on every sampled row at full capacity, including rows whose byte offset exceeds
    4 GiB and the rows written by a wrapping insert; the fused gather with (mean 0, var 1 - eps) is the plain gather."""
    from pql_amd import _lib as L
    from pql_amd.replay.simple_replay import ReplayBuffer
    rb = ReplayBuffer(cap, (O,), A, device=dev)
    chunk = 250_000
    first = cap - 3 * 4096 + 100            # leave the ring 12188 rows short of full ...
    def put(lo, hi):
        idx = torch.arange(lo, hi, device=dev)
        rb.add_to_buffer((_rows(idx, O, 1), _rows(idx, A, 2), _rows(idx, 1, 3), _rows(idx, O, 4), (_rows(idx, 1, 5) > 0.8).float()))
    for lo in range(0, first, chunk):
        put(lo, min(lo + chunk, first))
    assert rb.next_p == first and not rb.if_full
    put(first, first + 4 * 4096)            # ... then a wrapping insert: global rows [first, first + 16384)
    assert rb.if_full and rb.cur_capacity == cap and rb.next_p == first + 4 * 4096 - cap
    # global row number held by ring slot s after the wrap
    wrapped = rb.next_p
    def global_row(slot):
        return torch.where(slot < wrapped, slot + cap, slot)
    g = torch.Generator(device=dev).manual_seed(5)
    idx = torch.randint(cap, (B,), device=dev, generator=g)
    idx[:64] = torch.arange(cap - 64, cap, device=dev)          # the last slots: byte offset > 4 GiB for the 5 M ring
    idx[64:128] = torch.arange(0, 64, device=dev)               # the wrapped head
    obs, act, rew, nobs, done = rb.sample_batch(B, indices=idx)
    gr = global_row(idx)
    assert torch.equal(obs, _rows(gr, O, 1)) and torch.equal(act, _rows(gr, A, 2)) and torch.equal(nobs, _rows(gr, O, 4))
    assert torch.equal(rew, _rows(gr, 1, 3)) and torch.equal(done, (_rows(gr, 1, 5) > 0.8).float())
    if rb.ring.rec_ld * 4 * cap > 2 ** 32:
        assert int(idx.max()) * rb.ring.rec_ld * 4 > 2 ** 32
    # fused gather: normalisation with (0, 1 - eps, eps) divides by sqrt(1) and is the identity; no clamp
    ld_sa, ld_o = L.ld(O + A), L.ld(O)
    f = dict(dtype=torch.float32, device=dev)
    x_sa, xn_sa, xn_o = torch.zeros((B, ld_sa), **f), torch.zeros((B, ld_sa), **f), torch.zeros((B, ld_o), **f)
    r2, d2 = torch.empty(B, **f), torch.empty(B, **f)
    mean, var = torch.zeros(O, **f), torch.full((O,), 1.0 - 2.0 ** -13, **f)
    L.check(L.lib.pqlk_replay_gather_fused(C.byref(rb.ring.desc), L.ptr(idx), B, L.ptr(mean), L.ptr(var), 2.0 ** -13, 0, L.ptr(x_sa), ld_sa,
                                           L.ptr(xn_sa), L.ptr(xn_o), ld_o, L.ptr(r2), L.ptr(d2), L.stream(dev)))
    assert torch.equal(x_sa[:, :O], obs) and torch.equal(x_sa[:, O:O + A], act) and torch.equal(xn_sa[:, :O], nobs)
    assert torch.equal(xn_o[:, :O], nobs) and torch.equal(r2, rew.view(-1)) and torch.equal(d2, done.view(-1))
    assert torch.all(x_sa[:, O + A:] == 0) and torch.all(xn_o[:, O:] == 0)


def test_replay_round_trip_hbm_scale_ring(dev):
    """cfg #5's record (1 KiB) in a ring that takes most of the card's HBM (up to 200 M rows = 205 GB; BASELINE configs[4] calls
    itself a 288-GB buffer stress): insert -> gather is the identity on rows placed around every power-of-two byte boundary
    from 2^32 up, at the very end of the ring and across the wrap, through the plain and the fused K-batch gather.  Only the rows
    checked are written -- the property is the 64-bit addressing, not the fill."""
    from pql_amd import _lib as L
    from pql_amd.replay.simple_replay import ReplayBuffer
    O, A = 108, 21
    rec_bytes = int(L.lib.pqlk_replay_rec_ld(O, A)) * 4
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info(dev)
    cap = min(200_000_000, int(free * 0.75) // rec_bytes)
    if cap * rec_bytes < 2 ** 36:
        pytest.skip(f"only {free / 2 ** 30:.0f} GiB free on the card")
    rb = ReplayBuffer(cap, (O,), A, device=dev)
    try:
        _hbm_scale_body(rb, cap, rec_bytes, O, A, dev)
    finally:   # give the 200 GB back even when an assertion keeps this frame alive in a traceback
        rb.ring.records.untyped_storage().resize_(0)
        torch.cuda.empty_cache()


def _hbm_scale_body(rb, cap, rec_bytes, O, A, dev):
    from pql_amd import _lib as L
    assert rb.ring.rec_ld * 4 == rec_bytes and rb.ring.records.numel() * 4 == cap * rec_bytes
    n = 4096
    starts = [0]
    e = 32
    while 2 ** e < cap * rec_bytes:
        starts.append(2 ** e // rec_bytes - n // 2)   # rows either side of the 2^e-byte boundary
        e += 1
    starts.append(cap - n // 2)                       # the last rows + a wrap to the head (overwrites half of the first block)
    def rows_of(lo):
        idx = torch.arange(lo, lo + n, device=dev)
        return idx, (_rows(idx, O, 1), _rows(idx, A, 2), _rows(idx, 1, 3), _rows(idx, O, 4), (_rows(idx, 1, 5) > 0.8).float())
    for lo in starts:
        rb.next_p = lo                                # (public state of the reference's class too: simple_replay.py:29-31)
        rb.add_to_buffer(rows_of(lo)[1])
    assert rb.if_full and rb.next_p == n // 2 and rb.cur_capacity == cap
    slots = torch.cat([torch.arange(lo, lo + n, device=dev) % cap for lo in starts[1:]] + [torch.arange(n // 2, n, device=dev)])
    gr = torch.where(slots < n // 2, slots + cap, slots)   # global row a slot holds (the wrapped head carries rows cap .. cap + n/2)
    assert int(slots.max()) * rec_bytes > 2 ** 36
    obs, act, rew, nobs, done = rb.sample_batch(slots.numel(), indices=slots)
    assert torch.equal(obs, _rows(gr, O, 1)) and torch.equal(act, _rows(gr, A, 2)) and torch.equal(nobs, _rows(gr, O, 4))
    assert torch.equal(rew, _rows(gr, 1, 3)) and torch.equal(done, (_rows(gr, 1, 5) > 0.8).float())
    # the learners' launch: fused gather (fast path: flags 3), identity normalisation
    B = slots.numel()
    ld_sa = L.ld(O + A)
    f = dict(dtype=torch.float32, device=dev)
    x_sa, xn_sa = torch.zeros((B, ld_sa), **f), torch.zeros((B, ld_sa), **f)
    r2, d2 = torch.empty(B, **f), torch.empty(B, **f)
    mean, var = torch.zeros(O, **f), torch.full((O,), 1.0 - 2.0 ** -13, **f)
    perm = torch.randperm(B, device=dev)
    L.check(L.lib.pqlk_replay_gather_fused(C.byref(rb.ring.desc), L.ptr(slots[perm].contiguous()), B, L.ptr(mean), L.ptr(var), 2.0 ** -13, 3,
                                           L.ptr(x_sa), ld_sa, L.ptr(xn_sa), None, 0, L.ptr(r2), L.ptr(d2), L.stream(dev)))
    assert torch.equal(x_sa[:, :O], obs[perm]) and torch.equal(x_sa[:, O:O + A], act[perm]) and torch.equal(xn_sa[:, :O], nobs[perm])
    assert torch.equal(r2, rew.view(-1)[perm]) and torch.equal(d2, done.view(-1)[perm])


def test_nstep_properties_cfg4_shape(dev):
    """16384 envs x obs 211 (cfg #4), n = 3: with no dones the emitted transition is (obs_t, act_t, sum gamma^j r_{t+j},
    next_obs_{t+2}, 0); a done in the window truncates the return there, selects that step's next_obs and raises done;
    n = 1 is the identity."""
    from pql_amd.replay.nstep_replay import NStepReplay
    N, O, A, T, n = 16384, 211, 20, 6, 3
    g = torch.Generator(device=dev).manual_seed(1)
    obs = torch.randn((N, T, O), device=dev, generator=g); act = torch.rand((N, T, A), device=dev, generator=g)
    nobs = torch.randn((N, T, O), device=dev, generator=g)
    rew = torch.rand((N, T, 1), device=dev, generator=g)
    done = torch.zeros((N, T, 1), device=dev)
    done[::7, 2, 0] = 1.0                                   # every 7th env terminates at t = 2
    out = NStepReplay((O,), A, N, n, device=dev).add_to_buffer(obs, act, rew, nobs, done)
    o, a, R, no, d = (t.view(T - n + 1, N, -1) for t in out)     # time-major blocks of N rows
    gam = [0.99 ** j for j in range(n)]
    for t in range(T - n + 1):
        assert torch.equal(o[t], obs[:, t]) and torch.equal(a[t], act[:, t])
        win_done = done[:, t:t + n, 0]
        any_done = win_done.any(dim=1)
        first = torch.argmax(win_done, dim=1)
        keep = torch.arange(n, device=dev).unsqueeze(0) <= torch.where(any_done, first, torch.full_like(first, n)).unsqueeze(1)
        want = sum(gam[j] * rew[:, t + j, 0] * keep[:, j] for j in range(n))
        torch.testing.assert_close(R[t][:, 0], want, rtol=1e-6, atol=1e-6)
        sel = torch.where(any_done, t + first, torch.full_like(first, t + n - 1))
        assert torch.equal(no[t], nobs[torch.arange(N, device=dev), sel])
        assert torch.equal(d[t][:, 0], any_done.float())
    one = NStepReplay((O,), A, N, 1, device=dev).add_to_buffer(obs, act, rew, nobs, done)
    for got, src in zip(one, (obs, act, rew, nobs, done)):   # nstep_replay.py:66-67: the inputs themselves
        assert got is src


def test_c51_projection_conserves_mass_at_batch_32768(dev):
    """Categorical projection (distl_util.py:4-20): every projected row keeps the mass of its source pmf, stays in [0, 1],
    and a terminal transition puts everything on the bins around clamp(r)."""
    from pql_amd.utils.distl_util import projection
    B, K, vmin, vmax = 32768, 51, -10.0, 10.0
    g = torch.Generator(device=dev).manual_seed(2)
    p = torch.softmax(torch.randn((B, K), device=dev, generator=g) * 3, dim=1)
    rew = torch.randn((B, 1), device=dev, generator=g) * 6          # many rewards beyond +-v
    done = (torch.rand((B, 1), device=dev, generator=g) < 0.3).float()
    z = torch.linspace(vmin, vmax, K, device=dev)
    proj = projection(next_dist=p, reward=rew, done=done, gamma=0.99 ** 3, v_min=vmin, v_max=vmax, num_atoms=K, support=z, device=dev)
    torch.testing.assert_close(proj.sum(1), p.sum(1), rtol=0, atol=2e-6)
    assert float(proj.min()) >= 0.0 and float(proj.max()) <= 1.0 + 1e-6
    term = done.view(-1) > 0
    b = (rew.view(-1).clamp(vmin, vmax) - vmin) / ((vmax - vmin) / (K - 1))
    lo, hi = torch.floor(b).long().clamp(0, K - 1), torch.ceil(b).long().clamp(0, K - 1)
    mass = proj.gather(1, lo.unsqueeze(1)).view(-1) + torch.where(hi != lo, proj.gather(1, hi.unsqueeze(1)).view(-1), torch.zeros_like(b))
    torch.testing.assert_close(mass[term], torch.ones_like(mass[term]), rtol=0, atol=2e-6)


def test_optimizer_identities_full_arena(dev):
    """clip + AdamW + Polyak over the [512,512,256] DoubleQ arena (cfg #2): a zero gradient leaves exactly the weight-decay
    step p (1 - lr wd) with m = v = 0; tau = 1 makes the target a copy of the new parameters, tau = 0 leaves it untouched."""
    from pql_amd import _lib as L
    from pql_amd.models.mlp import ArenaLayout
    n = ArenaLayout([104, 512, 512, 256, 1], 2).total
    g = torch.Generator(device=dev).manual_seed(3)
    p0 = torch.randn(n, device=dev, generator=g) * 0.05
    t0 = torch.randn(n, device=dev, generator=g) * 0.05
    for tau in (1.0, 0.0):
        p, tg = p0.clone(), t0.clone()
        m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        step = torch.zeros(1, dtype=torch.int32, device=dev); scr = torch.zeros(2048, device=dev); gn = torch.zeros(1, device=dev)
        L.check(L.lib.pqlk_clip_adamw_polyak(L.ptr(p), L.ptr(torch.zeros(n, device=dev)), L.ptr(m), L.ptr(v), L.ptr(tg), n, 1.0, 0.5, 5e-4,
                                             0.9, 0.999, 1e-8, 1e-2, tau, L.ptr(step), L.ptr(gn), L.ptr(scr), L.stream(dev)))
        torch.cuda.synchronize()
        assert int(step) == 1 and float(gn) == 0.0
        assert torch.equal(p, p0 * np.float32(1 - 5e-4 * 1e-2)) and not m.any() and not v.any()
        assert torch.equal(tg, p) if tau == 1.0 else torch.equal(tg, t0)


def test_mlp_backward_is_linear_and_forward_deterministic_at_batch_32768(dev):
    """cfg #5 shapes (obs 108, act 21, batch 32768, [512,512,256] twin critic): two forwards are bitwise equal (fused and
    per-layer alike), and the parameter / input gradients are linear in dy to fp32 rounding."""
    from pql_amd import _lib as L
    from pql_amd.models.mlp import ArenaLayout, PackedWeights, default_splits, mlp_forward_raw
    O, A, B = 108, 21, 32768
    lay = ArenaLayout([O + A, 512, 512, 256, 1], 2)
    g = torch.Generator(device=dev).manual_seed(4)
    arena = torch.zeros(lay.total, device=dev)
    for net in range(2):
        for l in range(lay.n_layers):
            bound = 1.0 / np.sqrt(lay.dims[l])
            lay.weight(arena, net, l).copy_((torch.rand((lay.dims[l + 1], lay.dims[l]), device=dev, generator=g) * 2 - 1) * bound)
            lay.bias(arena, net, l).copy_((torch.rand((lay.dims[l + 1],), device=dev, generator=g) * 2 - 1) * bound)
    x = torch.zeros((B, lay.ld_in), device=dev)
    x[:, : O + A] = torch.randn((B, O + A), device=dev, generator=g)
    pk = PackedWeights(lay, dev).refresh(arena)
    a1 = mlp_forward_raw(lay, arena, x, L.ACT_NONE, packed=pk, stash_all=True)
    a2 = mlp_forward_raw(lay, arena, x, L.ACT_NONE, packed=pk, stash_all=True)
    a3 = mlp_forward_raw(lay, arena, x, L.ACT_NONE)
    assert torch.equal(a1, a2)
    n_out = 2 * B * lay.ld_out                       # the output block sits at the end of the stash
    assert torch.equal(a1[:-n_out], a3[:-n_out])     # hidden layers: fused == per-layer, bitwise
    torch.testing.assert_close(a1[-n_out:], a3[-n_out:], rtol=2e-6, atol=2e-6)   # fused output layer: split reduction (reassociation)
    splits = default_splits(B)
    ws = torch.empty(lay.bwd_ws_floats(B, splits), device=dev)

    def grads(dy):
        gr, dx = torch.empty_like(arena), torch.empty((B, lay.ld_in), device=dev)
        L.check(L.lib.pqlk_mlp_backward(C.byref(lay.desc), L.ptr(arena), L.ptr(x), lay.ld_in, B, L.ptr(a1), L.ptr(dy), L.ptr(gr), splits,
                                        L.ptr(dx), lay.ld_in, 0, 0, None, 0, L.ptr(ws), ws.numel(), L.stream(dev)))
        return gr, dx

    dy1 = torch.zeros((2, B, lay.ld_out), device=dev); dy2 = torch.zeros_like(dy1)
    dy1[:, :, 0] = torch.randn((2, B), device=dev, generator=g) / B
    dy2[:, :, 0] = torch.randn((2, B), device=dev, generator=g) / B
    (g1, x1), (g2, x2), (g12, x12) = grads(dy1), grads(dy2), grads(dy1 + 2.0 * dy2)
    scale_g, scale_x = float(g12.abs().max()), float(x12.abs().max())
    assert float((g1 + 2.0 * g2 - g12).abs().max()) <= 2e-5 * scale_g
    assert float((x1 + 2.0 * x2 - x12).abs().max()) <= 2e-5 * scale_x
    g1b, _ = grads(dy1)
    assert torch.equal(g1, g1b)   # split-batch slabs are reduced in a fixed order: bitwise reproducible
