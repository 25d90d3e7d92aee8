"""Data-parallel path on CPU with the gloo backend, world_size 2 (the N > 1 layout of bench.py /
scripts/train_pql.py: env axis + replay shard per rank, ONE gradient all-reduce per step, replicated optimiser).

What runs here is the host-side and collective logic, on CPU tensors:
  * shard gradients all-reduced and scaled 1/world == the full-batch gradient (SURVEY 8e parity rule),
    followed by the replicated optimiser step keeping replicas identical;
  * RunningMeanStd.merge_batch across ranks == single-process merge in rank order;
  * the synthetic env and the mixed exploration noise index the GLOBAL env axis, so shards reproduce slices.
The HIP kernels themselves are covered by the -m gpu tests; nothing here launches one."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import detdata as dd

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def run_world(fn, world):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def run2(fn):
    return run_world(fn, 2)


# --------------------------------------------------------------------------- gradient all-reduce == full batch
def _dp_grad(rank, world):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import pql_ref_cpu as ref
    O, A, B = 8, 2, 64
    st = dd.doubleq_state(O, A, 1, 21)
    q1 = [p.requires_grad_(True) for p in ref.params_from_state(st, "net_q1.net.")]
    q2 = [p.requires_grad_(True) for p in ref.params_from_state(st, "net_q2.net.")]
    obs, act, tgt = T(dd.uniform((B, O), 1, -2, 2)), T(dd.uniform((B, A), 2)), T(dd.uniform((B, 1), 3))
    sl = slice(rank * B // world, (rank + 1) * B // world)          # rank r takes idx[r*B/G:(r+1)*B/G]
    a, b = ref.twin_forward_ref(q1, q2, obs[sl], act[sl])
    loss = torch.nn.functional.mse_loss(a, tgt[sl]) + torch.nn.functional.mse_loss(b, tgt[sl])
    grads = torch.autograd.grad(loss, [*q1, *q2])
    flat = torch.cat([g.reshape(-1) for g in grads])                # the flat gradient arena
    dist.all_reduce(flat)                                           # the ONE collective of a DP step (sum)
    flat *= 1.0 / world                                             # folded into pqlk_clip_adamw_polyak(grad_scale)
    opt = ref.AdamWRef([p.detach().clone() for p in (*q1, *q2)])
    off, gl = 0, []
    for p in opt.params:
        gl.append(flat[off: off + p.numel()].view_as(p)); off += p.numel()
    opt.apply(gl, 0.5)
    return flat.numpy(), torch.cat([p.reshape(-1) for p in opt.params]).numpy()


def test_allreduced_shard_gradients_equal_full_batch_gradient():
    from oracle import pql_ref_cpu as ref
    (g0, p0), (g1, p1) = run2(_dp_grad)
    assert np.array_equal(g0, g1) and np.array_equal(p0, p1)        # replicas stay bit-identical
    O, A, B = 8, 2, 64
    st = dd.doubleq_state(O, A, 1, 21)
    q1 = [p.requires_grad_(True) for p in ref.params_from_state(st, "net_q1.net.")]
    q2 = [p.requires_grad_(True) for p in ref.params_from_state(st, "net_q2.net.")]
    obs, act, tgt = T(dd.uniform((B, O), 1, -2, 2)), T(dd.uniform((B, A), 2)), T(dd.uniform((B, 1), 3))
    a, b = ref.twin_forward_ref(q1, q2, obs, act)
    loss = torch.nn.functional.mse_loss(a, tgt) + torch.nn.functional.mse_loss(b, tgt)
    full = torch.cat([g.reshape(-1) for g in torch.autograd.grad(loss, [*q1, *q2])]).numpy()
    np.testing.assert_allclose(g0, full, rtol=1e-5, atol=1e-8)      # 1e-6..1e-5 relative


# --------------------------------------------------------------------------- running statistics
def _rms(rank, world):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pql_amd.utils.torch_util import RunningMeanStd
    rms = RunningMeanStd(shape=(6,), device="cpu")
    rms.pg = dist.group.WORLD
    for step in range(3):
        x = T(dd.uniform((32, 6), 100 + 10 * step + rank, -3, 5))   # this rank's env shard
        rms.merge_batch(x.mean(0), x.var(0), x.shape[0])            # (batch moments come from a HIP launch on GPU)
    return np.concatenate([rms.mean.numpy(), rms.var.numpy(), [rms.count]])


def test_running_mean_std_merges_identically_on_every_rank():
    from oracle import pql_ref_cpu as ref
    a, b = run2(_rms)
    assert np.array_equal(a, b)
    single = ref.RunningMeanStdRef((6,))
    for step in range(3):
        for rank in range(2):
            single.update(T(dd.uniform((32, 6), 100 + 10 * step + rank, -3, 5)))
    np.testing.assert_allclose(a, np.concatenate([single.mean.numpy(), single.var.numpy(), [single.count]]), rtol=1e-6)


# --------------------------------------------------------------------------- env / noise sharding (no process group needed)
def test_env_shards_reproduce_slices_of_the_global_env():
    from pql_amd.envs.synthetic import SyntheticVecEnv
    full = SyntheticVecEnv(64, 5, 3, device="cpu", seed=42)
    shards = [SyntheticVecEnv(32, 5, 3, device="cpu", seed=42, env_offset=off) for off in (0, 32)]
    o = full.reset(); os_ = [s.reset() for s in shards]
    assert torch.equal(o, torch.cat(os_))
    act = T(dd.uniform((64, 3), 5))
    for _ in range(3):
        n, r, d, info = full.step(act)
        parts = [s.step(act[i * 32:(i + 1) * 32]) for i, s in enumerate(shards)]
        assert torch.equal(n, torch.cat([p[0] for p in parts]))
        assert torch.equal(r, torch.cat([p[1] for p in parts])) and torch.equal(d, torch.cat([p[2] for p in parts]))
        assert not info["TimeLimit.truncated"].any()
    assert abs(float(o.mean())) < 0.2 and 0.8 < float(o.std()) < 1.2      # N(0,1) observations


def test_mixed_noise_uses_global_env_index():
    from pql_amd.utils.noise import add_mixed_normal_noise
    std = torch.linspace(0.05, 0.8, 8)                                       # sigma of GLOBAL env e = linspace(...)[e]
    for off in (0, 4):
        torch.manual_seed(1); got = add_mixed_normal_noise(torch.zeros(4, 2), 0.8, 0.05, env_offset=off, total_envs=8)
        torch.manual_seed(1); draw = torch.empty(4, 2).normal_()
        assert torch.allclose(got, draw * std[off: off + 4].unsqueeze(1))
    torch.manual_seed(2); whole = add_mixed_normal_noise(torch.zeros(8, 2), 0.8, 0.05)
    torch.manual_seed(2); draw = torch.empty(8, 2).normal_()
    assert torch.allclose(whole, draw * std.unsqueeze(1))                    # single-GPU form unchanged


# --------------------------------------------------------------------------- shard sizes (strong / weak) and communicators
def test_shard_sizes_strong_and_weak():
    from pql_amd.utils.dp import shard
    # BASELINE configs[3] as written: 16384 envs, replay 5 M, batch 8192 over 8 GPUs
    parts = [shard(16384, 5_000_000, 8192, 8, r, "strong") for r in range(8)]
    assert all((p.num_envs, p.memory_size, p.batch_size, p.total_envs) == (2048, 625_000, 1024, 16384) for p in parts)
    assert [p.env_offset for p in parts] == [2048 * r for r in range(8)] and parts[0].global_batch == 8192
    w = shard(4096, 1_000_000, 8192, 4, 3, "weak")
    assert (w.num_envs, w.memory_size, w.batch_size, w.total_envs, w.env_offset, w.global_batch) == (4096, 1_000_000, 8192, 16384, 12288, 32768)
    one = shard(4096, 1_000_000, 8192, 1, 0, "strong")
    assert (one.num_envs, one.batch_size, one.scaling) == (4096, 8192, "single")
    with pytest.raises(ValueError, match="num_envs"):
        shard(4097, 1_000_000, 8192, 2, 0, "strong")
    with pytest.raises(ValueError, match="batch_size"):
        shard(4096, 1_000_000, 8191, 2, 0, "strong")


def _groups(rank, world):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pql_amd.utils.dp import component_groups
    g = component_groups(dist.group.WORLD)
    assert len({id(x) for x in g.values()}) == 3 and all(dist.get_world_size(x) == world for x in g.values())
    # collectives issued in a DIFFERENT order per group on the two ranks still pair up: the groups are independent queues
    order = ("v", "p", "rms") if rank == 0 else ("rms", "p", "v")
    vals = {n: torch.full((4,), float(10 * i + rank + 1)) for i, n in enumerate(("v", "p", "rms"))}
    works = [dist.all_reduce(vals[n], group=g[n], async_op=True) for n in order]
    for w in works:
        w.wait()
    return np.stack([vals[n].numpy() for n in ("v", "p", "rms")])


def test_component_groups_are_independent_communicators():
    a, b = run2(_groups)
    want = np.stack([np.full(4, 20.0 * i + 3.0) for i in range(3)])   # (10 i + 1) + (10 i + 2)
    assert np.array_equal(a, want) and np.array_equal(b, want)


def _strong_half_steps(rank, world):
    """Two ranks in STRONG mode: each owns half the envs' rows of a sharded ring and half the batch; the all-reduced mean
    gradient and the replicated optimiser step must reproduce the one-process step on the whole batch."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import pql_ref_cpu as ref
    from pql_amd.utils.dp import shard
    O, A, B, rows = 8, 2, 64, 96
    sh = shard(32, 2 * rows, B, world, rank, "strong")
    assert (sh.memory_size, sh.batch_size) == (rows, B // 2)
    cst, ast = dd.doubleq_state(O, A, 1, 21), dd.mlp_state(O, A, 11)
    hp = ref.HyperRef(batch_size=sh.batch_size)
    v = ref.VLearnerRef(O, A, hp, sh.memory_size, ref.params_from_state(cst, "net_q1.net."), ref.params_from_state(cst, "net_q2.net."))
    whole = (T(dd.uniform((2 * rows, O), 1, -3, 3)), T(dd.uniform((2 * rows, A), 2)), T(dd.uniform((2 * rows, 1), 3, -0.05, 0.05)),
             T(dd.uniform((2 * rows, O), 4, -3, 3)), T(dd.bernoulli((2 * rows, 1), 5, 0.1)))
    mine = tuple(t[rank * rows:(rank + 1) * rows] for t in whole)          # this rank's envs' transitions -> its ring shard
    mean, var = T(dd.uniform((O,), 6, -0.5, 0.5)), T(dd.uniform((O,), 7, 0.5, 2.0))
    v.update(ref.params_from_state(ast), mine, (mean, var, 1e-4))
    idx_local = T(dd.integers((sh.batch_size,), 8 + rank, rows))           # B/G uniform draws from the shard
    draw = T(dd.uniform((B, A), 9, -2, 2))[rank * sh.batch_size:(rank + 1) * sh.batch_size]
    loss, grads = v.loss_and_grads(idx=idx_local, draw=draw)
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat)
    flat *= 1.0 / world
    return flat.numpy(), float(loss.detach())


def test_strong_scaling_two_ranks_reproduce_the_one_process_gradient():
    from oracle import pql_ref_cpu as ref
    if not hasattr(ref.VLearnerRef, "loss_and_grads"):
        pytest.skip("oracle has no loss_and_grads")
    (g0, l0), (g1, l1) = run2(_strong_half_steps)
    assert np.array_equal(g0, g1)
    O, A, B, rows = 8, 2, 64, 96
    cst, ast = dd.doubleq_state(O, A, 1, 21), dd.mlp_state(O, A, 11)
    hp = ref.HyperRef(batch_size=B)
    v = ref.VLearnerRef(O, A, hp, 2 * rows, ref.params_from_state(cst, "net_q1.net."), ref.params_from_state(cst, "net_q2.net."))
    whole = (T(dd.uniform((2 * rows, O), 1, -3, 3)), T(dd.uniform((2 * rows, A), 2)), T(dd.uniform((2 * rows, 1), 3, -0.05, 0.05)),
             T(dd.uniform((2 * rows, O), 4, -3, 3)), T(dd.bernoulli((2 * rows, 1), 5, 0.1)))
    mean, var = T(dd.uniform((O,), 6, -0.5, 0.5)), T(dd.uniform((O,), 7, 0.5, 2.0))
    v.update(ref.params_from_state(ast), whole, (mean, var, 1e-4))
    # the same samples in the one-process ring: rank r's local row i is global row r * rows + i
    idx = torch.cat([T(dd.integers((B // 2,), 8 + r, rows)) + r * rows for r in range(2)])
    loss, grads = v.loss_and_grads(idx=idx, draw=T(dd.uniform((B, A), 9, -2, 2)))
    full = torch.cat([g.reshape(-1) for g in grads]).numpy()
    np.testing.assert_allclose(g0, full, rtol=1e-5, atol=1e-9)
    assert abs(0.5 * (l0 + l1) - float(loss.detach())) <= 1e-6 * max(1.0, abs(float(loss.detach())))   # mean of rank means == global batch mean


# --------------------------------------------------------------------------- gradient buckets (all-reduce behind backward)
def _buckets(rank, world):
    """The per-layer buckets tile a twin critic's gradient arena exactly once, and their grouped all-reduces leave what one
    all-reduce of the whole arena leaves (CPU tensors through the same BucketAllReduce the learner uses)."""
    from pql_amd.models.mlp import ArenaLayout
    from pql_amd.utils.dp import BucketAllReduce, bucket_views, layer_buckets
    lay = ArenaLayout([104, 512, 512, 256, 1], 2)     # configs[1]: obs 88 + act 16
    bk = layer_buckets(lay.n_layers)
    assert bk == [(3, 2), (1, 1), (0, 0)] and layer_buckets(2) == [(1, 0)] and layer_buckets(5) == [(4, 3), (2, 2), (1, 1), (0, 0)]
    g = T(dd.uniform((lay.total,), 40 + rank, -1, 1)).clone()
    cover = torch.zeros(lay.total, dtype=torch.int32)
    for hi, lo in bk:
        for v, c in zip(bucket_views(g, lay, hi, lo), bucket_views(cover, lay, hi, lo)):
            assert v.is_contiguous() and v.storage_offset() % 32 == 0 and v.numel() % 32 == 0   # 128-B aligned ranges of the arena
            c += 1
    assert bool((cover == 1).all())                   # every arena element in exactly one bucket
    for net in range(lay.n_nets):                     # ... and the head's block sits in the first one, with the last hidden layer
        v = bucket_views(g, lay, *bk[0])[net]
        assert v.numel() == lay.net_stride - lay.w_off[2]
    whole = g.clone()
    dist.all_reduce(whole)
    red = BucketAllReduce(dist.group.WORLD)
    for hi, lo in bk:
        red.issue(bucket_views(g, lay, hi, lo))
    red.wait()
    return np.array_equal(g.numpy(), whole.numpy()), float(whole.abs().sum())


def test_gradient_buckets_tile_the_arena_and_sum_like_one_all_reduce():
    a, b = run2(_buckets)
    assert a[0] and b[0] and a[1] == b[1] and a[1] > 0


# --------------------------------------------------------------------------- BASELINE configs[3] as written: EIGHT ranks
CFG3 = dict(num_envs=16384, memory_size=5_000_000, batch=8192, O=211, A=20, atoms=51, hidden=[512, 512, 256])


def _cfg3_rank(rank, world):
    """One of the 8 ranks of configs[3] (PQL-D, 16384 ShadowHand-shape envs, replay 5 M, batch 8192, env-sharded + gradient
    all-reduce): the host-side and collective logic of the data-parallel path at the sizes the config names -- shard arithmetic,
    the rank's env shard, the merged running statistics, the global noise index, the gradient buckets of the C51 twin critic."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pql_amd.envs.synthetic import SyntheticVecEnv
    from pql_amd.models.mlp import ArenaLayout
    from pql_amd.utils.dp import BucketAllReduce, bucket_views, component_groups, layer_buckets, shard
    from pql_amd.utils.noise import add_mixed_normal_noise
    from pql_amd.utils.torch_util import RunningMeanStd
    c = CFG3
    sh = shard(c["num_envs"], c["memory_size"], c["batch"], world, rank, "strong")
    out = dict(shard=(sh.num_envs, sh.memory_size, sh.batch_size, sh.total_envs, sh.env_offset, sh.global_batch))
    groups = component_groups(dist.group.WORLD)              # three communicators, as scripts/train_pql.py makes them
    # the rank's env shard: first observations + two steps (checked against slices of the 16384-env job by the parent)
    env = SyntheticVecEnv(sh.num_envs, c["O"], c["A"], device="cpu", seed=42, env_offset=sh.env_offset)
    obs = env.reset()
    out["obs_sum"] = float(obs.double().sum()); out["obs_head"] = obs[:2, :3].numpy().copy()
    # running statistics: every rank folds in EVERY rank's batch moments in rank order (torch_util.py:91-103 per batch)
    rms = RunningMeanStd(shape=(c["O"],), device="cpu")
    rms.pg = groups["rms"]
    x = obs
    for step in range(2):
        rms.merge_batch(x.mean(0), x.var(0), x.shape[0])
        x, _, _, _ = env.step(T(dd.uniform((sh.num_envs, c["A"]), 900 + step)))
    out["rms"] = np.concatenate([rms.mean.numpy(), rms.var.numpy(), [rms.count]])
    # exploration noise: sigma of GLOBAL env e = linspace(std_min, std_max, 16384)[e]  (noise.py:30-41)
    torch.manual_seed(5)
    noise = add_mixed_normal_noise(torch.zeros(sh.num_envs, c["A"]), 0.8, 0.05, env_offset=sh.env_offset, total_envs=sh.total_envs)
    torch.manual_seed(5)
    draw = torch.empty(sh.num_envs, c["A"]).normal_()
    std = torch.linspace(0.05, 0.8, sh.total_envs)[sh.env_offset: sh.env_offset + sh.num_envs]
    out["noise_ok"] = bool(torch.allclose(noise, draw * std.unsqueeze(1)))
    # the C51 twin critic's gradient arena in per-layer buckets on the V-learner's communicator == one all-reduce of the arena
    lay = ArenaLayout([c["O"] + c["A"], *c["hidden"], c["atoms"]], 2)
    g = T(dd.uniform((lay.total,), 70 + rank, -1, 1)).clone()
    whole = g.clone()
    dist.all_reduce(whole, group=groups["v"])
    cover = torch.zeros(lay.total, dtype=torch.int32)
    red = BucketAllReduce(groups["v"])
    for hi, lo in layer_buckets(lay.n_layers):
        red.issue(bucket_views(g, lay, hi, lo))
        for cv in bucket_views(cover, lay, hi, lo):
            cv += 1
    red.wait()
    # every element in exactly one bucket; the bucketed sums are the one-collective sums up to the ORDER the ranks' addends are taken
    # in (a ring all-reduce starts each chunk at a different rank, and the chunking follows the message size: with more than two
    # ranks the two forms differ in the last bit of some elements, while every RANK still receives the same bits -- checked below)
    out["tiling_ok"] = bool((cover == 1).all())
    out["bucket_err"] = float((g - whole).abs().max() / whole.abs().max())
    out["grad_fp"] = float(whole.double().abs().sum())
    out["bucket_fp"] = float(g.double().sum())
    # per-rank batch means average to the job's batch mean (equal shards): all-reduce(sum) / G of a per-rank scalar
    m = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(m, group=groups["p"])
    out["mean_of_means"] = float(m / world)
    return out


def test_configs3_split_as_written_on_eight_ranks():
    """BASELINE configs[3] ("16384 envs ... 8 x MI355X env-sharded + RCCL grad all-reduce") split over EIGHT gloo ranks exactly as
    scripts/train_pql.py / bench.py --scaling strong would split it: 2048 envs / 625 000 rows / batch 1024 per rank.  All 8 ranks
    must hold identical running statistics, equal to the single-process merge in rank order; the shards are slices of the job's env
    axis; the noise scale follows the global env index; the per-layer gradient buckets tile the C51 critic's arena and sum like one
    all-reduce.  (The reference has no data parallelism: scripts/train_pql.py:41-70 places three Ray actors on GPUs by function.)"""
    from oracle import pql_ref_cpu as ref
    from pql_amd.envs.synthetic import SyntheticVecEnv
    world, c = 8, CFG3
    outs = run_world(_cfg3_rank, world)
    assert [o["shard"] for o in outs] == [(2048, 625_000, 1024, 16384, 2048 * r, 8192) for r in range(world)]
    for o in outs[1:]:
        assert np.array_equal(o["rms"], outs[0]["rms"])                       # identical on all 8 ranks, bit for bit
        assert o["grad_fp"] == outs[0]["grad_fp"] and o["bucket_fp"] == outs[0]["bucket_fp"]   # replicas stay identical either way
    assert all(o["noise_ok"] and o["tiling_ok"] for o in outs)
    assert all(o["bucket_err"] < 1e-6 for o in outs)
    assert all(abs(o["mean_of_means"] - 4.5) < 1e-12 for o in outs)
    # the job's env in ONE process: shards are slices, and the merged statistics are the rank-order merge of the slices' moments
    full = SyntheticVecEnv(c["num_envs"], c["O"], c["A"], device="cpu", seed=42)
    x = full.reset()
    for r, o in enumerate(outs):
        sl = x[2048 * r: 2048 * (r + 1)]
        assert np.array_equal(o["obs_head"], sl[:2, :3].numpy()) and abs(o["obs_sum"] - float(sl.double().sum())) < 1e-6
    single = ref.RunningMeanStdRef((c["O"],))
    for step in range(2):
        for r in range(world):
            single.update(x[2048 * r: 2048 * (r + 1)])
        x, _, _, _ = full.step(torch.cat([T(dd.uniform((2048, c["A"]), 900 + step)) for _ in range(world)]))
    np.testing.assert_allclose(outs[0]["rms"], np.concatenate([single.mean.numpy(), single.var.numpy(), [single.count]]), rtol=2e-6, atol=1e-7)
