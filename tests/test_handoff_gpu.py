"""The measured mode (learners on their own HIP streams, steps replayed from hipGraphs) and the two-GPU hand-off path,
checked against the serial path.  `bench.py` reports `value` in streams + graph mode, so that exact schedule object is what
runs here.  Run with `pytest -m gpu`.

What is asserted
  * bit-identity: N slices of the 1 : 4 : 8 schedule (rollout -> n-step -> ring inserts -> weight hand-offs -> V / P steps)
    give the SAME parameter arenas, optimiser state, replay rings and observation statistics whether every launch sits on
    one stream (no graph) or V-learner / P-learner / rollout run on three streams with hipGraph replay -- and also when every
    hand-off is forced through the copy streams + landing blocks of the two-GPU layout (both "devices" = cuda:0);
  * no torn hand-off: with the learners free-running in threads, every critic snapshot a consumer reads equals the live
    arena at a whole optimiser step (checksum taken on the owner's stream at publish time);
  * the reference's topology (threads + ratio controller) keeps the update counts near 1 : P : V = 1 : 4 : 8.
"""
import argparse
import importlib.util
import os
import sys
import threading
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _args(**kw):
    a = dict(task="Toy", num_envs=64, batch=256, replay=4096, nstep=3, hidden="512,256,128", distl=False, no_graph=False,
             no_streams=False, no_fused=False)
    a.update(kw)
    return argparse.Namespace(**a)


def _state(actor, v, p):
    torch.cuda.synchronize()
    return dict(critic=v.critic.arena.data.clone(), target=v.critic_target.arena.data.clone(), adam_m=v.opt.m.clone(),
                adam_v=v.opt.v.clone(), v_step=v.opt.step.clone(), policy=p.actor.arena.data.clone(), p_m=p.opt.m.clone(),
                p_step=p.opt.step.clone(), v_ring=v.memory.ring.records.clone(), p_ring=p.ring.records.clone(),
                v_loss=v.loss_ring.clone(), p_loss=p.loss_ring.clone(), rollout_policy=actor.actor.arena.data.clone(),
                rms_mean=actor.obs_rms.mean.clone(), rms_var=actor.obs_rms.var.clone(), obs=actor.obs.clone(),
                v_replica_of_policy=v.actor.arena.data.clone(), p_replica_of_critic=p.critic.arena.data.clone())


def _run_schedule(steps, force_ship=False, **kw):
    from pql_amd.utils import handoff
    bench = _bench()
    handoff.FORCE_SHIP = force_ship
    try:
        dev = torch.device("cuda:0")
        torch.manual_seed(1234)
        args = _args(**kw)
        cfg, env, actor, v, p = bench.build_system(args, 0, 1, dev, None)
        critic, policy = bench.prefill(actor, v, p, env, cfg, args, dev)
        sched = bench.Schedule(actor, v, p, env, cfg, dev, critic, policy)
        for _ in range(steps):
            sched.step()
        out = _state(actor, v, p)
        out["counts"] = (v.update_count, p.update_count, sched.global_steps)
        return out
    finally:
        handoff.FORCE_SHIP = False


@pytest.mark.parametrize("distl", [False, True])
def test_streams_and_graphs_are_bit_identical_to_the_serial_schedule(distl):
    """The mode BENCH is measured in vs the plain one-stream eager path, 48 slices = 6 rollout iterations with hand-offs."""
    serial = _run_schedule(48, no_graph=True, no_streams=True, distl=distl)
    fast = _run_schedule(48, distl=distl)
    assert serial["counts"] == fast["counts"] == (48, 24, 6 * 64)
    assert int(serial["v_step"]) == 48 and int(serial["p_step"]) == 24
    for k, a in serial.items():
        if k != "counts":
            assert torch.equal(a, fast[k]), k
    captured_rng = _run_schedule(48, distl=distl, graph_rng=True)   # the RNG draws inside the graphs instead of in front of them
    for k, a in serial.items():
        if k != "counts":
            assert torch.equal(a, captured_rng[k]), k
    # and the hand-offs did move: the rollout replica holds a policy that has been stepped
    assert not torch.equal(fast["rollout_policy"], _run_schedule(0)["rollout_policy"])


def test_two_gpu_hand_off_path_rehearsed_on_one_card_is_bit_identical():
    """cfg #3 (simulator GPU0, learners GPU1) moves every block through a Shipper: copy streams, landing blocks, lease
    events.  With both devices = cuda:0 the same code must reproduce the serial arenas bit for bit."""
    serial = _run_schedule(40, no_graph=True, no_streams=True)
    shipped = _run_schedule(40, force_ship=True)
    for k, a in serial.items():
        if k != "counts":
            assert torch.equal(a, shipped[k]), k
    eager_streams = _run_schedule(40, force_ship=True, no_graph=True)
    for k, a in serial.items():
        if k != "counts":
            assert torch.equal(a, eager_streams[k]), k


def test_full_size_schedule_streams_graph_vs_serial():
    """Same bit-identity at BASELINE configs[1] shapes (obs 88 / act 16, batch 8192, [512,512,256]), smaller ring."""
    kw = dict(task="AllegroHand", num_envs=4096, batch=8192, replay=65536, hidden="512,512,256")
    serial = _run_schedule(24, no_graph=True, no_streams=True, **kw)
    fast = _run_schedule(24, **kw)
    for k, a in serial.items():
        if k != "counts":
            assert torch.equal(a, fast[k]), k


@pytest.mark.parametrize("force_ship", [False, True])
def test_free_running_learners_never_hand_out_a_torn_arena(force_ship):
    """Learner threads step continuously while the main thread keeps handing data over.  Every snapshot returned by
    `update()` must equal the live arena as it was between two optimiser steps: the owner records a checksum of the live
    arena right behind the snapshot copy (same stream), the consumer checksums the snapshot behind its acquire fence."""
    from pql_amd.algo.pql_p_learner import asyn_p_learner
    from pql_amd.algo.pql_v_learner import asyn_v_learner
    from pql_amd.utils import handoff as H
    bench = _bench()
    H.FORCE_SHIP = force_ship
    stop = threading.Event()
    threads = []
    try:
        dev = torch.device("cuda:0")
        torch.manual_seed(7)
        args = _args(task="AllegroHand", num_envs=256, batch=2048, replay=16384)
        cfg, env, actor, v, p = bench.build_system(args, 0, 1, dev, None)
        critic, policy = bench.prefill(actor, v, p, env, cfg, args, dev)
        owner_sums = {"v": [], "p": []}
        for tag, learner, live in (("v", v, v.critic), ("p", p, p.actor)):
            def wrap(orig=learner._pub.publish, live=live, tag=tag):
                snap = orig()
                owner_sums[tag].append(live.arena.data.double().sum())   # current stream = the owner's, right behind the copy
                snap._test_index = len(owner_sums[tag]) - 1
                return snap
            learner._pub.publish = wrap
        v.use_private_rng(11); p.use_private_rng(12)
        threads = [threading.Thread(target=asyn_v_learner, args=(v, cfg, stop), daemon=True),
                   threading.Thread(target=asyn_p_learner, args=(p, cfg, stop), daemon=True)]
        for t in threads:
            t.start()
        checker = torch.cuda.Stream(dev)
        seen = []
        rms = actor.obs_rms
        for it in range(60):
            p_data, v_data, _ = actor.explore_env(env, 1, random=False)
            critic, _, n_v = v.update(policy, v_data, rms.get_states(dev), 0)
            policy, _, n_p = p.update(critic, p_data, rms.get_states(dev), 0)
            actor.set_actor(policy)
            for tag, snap in (("v", critic), ("p", policy)):
                with H.LOCK:
                    lease = H.acquire(snap, checker)
                    with torch.cuda.stream(checker):
                        seen.append((tag, snap._test_index, snap.arena.data.double().sum()))
                    H.release(lease, checker)
            time.sleep(0.001)
        stop.set()
        for t in threads:
            t.join(timeout=60)
        torch.cuda.synchronize()
        assert v.update_count > 20 and p.update_count > 10, (v.update_count, p.update_count)
        assert len(seen) == 120
        for tag, i, s in seen:
            assert s.item() == owner_sums[tag][i].item(), (tag, i)
        # the learners did step between hand-offs (the checksums are not all one value)
        assert len({owner_sums["v"][i].item() for t, i, _ in seen if t == "v"}) > 5
        assert torch.isfinite(v.critic.arena.data).all() and torch.isfinite(p.actor.arena.data).all()
    finally:
        stop.set()
        for t in threads:
            t.join(timeout=60)
        H.FORCE_SHIP = False


def test_train_pql_free_running_with_ratio_controller():
    """scripts/train_pql.py with algo.async_learners=True: threads + the reference's controller.  The control law needs
    thousands of iterations to settle exactly (tests/test_host_cpu.py runs it closed-loop in virtual time); here the real
    threads must run, stop cleanly and land near the design ratios."""
    from pql_amd.utils.cfg import load_cfg
    spec = importlib.util.spec_from_file_location("train_pql", os.path.join(ROOT, "scripts", "train_pql.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cfg = load_cfg(["task=AllegroHand", "num_envs=1024", "algo.batch_size=4096", "algo.memory_size=200000", "algo.num_gpus=1",
                    "max_step=1500000", "algo.async_learners=True", "algo.eval_freq=100000", "algo.log_freq=100000"])
    out = mod.main(cfg)
    iters = out["rollout_iterations"]
    assert iters > 1000
    v_per, p_per = out["critic_updates"] / iters, out["actor_updates"] / iters
    assert 4.0 < v_per < 12.0, out
    assert 1.5 < out["critic_updates"] / out["actor_updates"] < 2.7, out
    assert p_per > 1.5, out


def _run_bench(*flags, timeout=600):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--num-envs", "256", "--batch", "1024", "--replay", "20000",
                        "--steps", "16", "--warmup", "8", "--repeat", "2", "--no-cpu-baseline", *flags],
                       env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks_for_data_parallel():
    """`python bench.py --gpus 2` from a bare shell (no torchrun, no WORLD_SIZE) must start two ranks itself.  One card here, so
    the rehearsal form: gloo + --share-gpu, and the line has to say so instead of passing for a two-GPU RCCL run."""
    line = _run_bench("--gpus", "2", "--backend", "gloo", "--share-gpu")
    cfg = line["config"]
    assert cfg["ranks"] == 2 and cfg["backend"] == "gloo" and cfg["share_gpu"] is True and cfg["parallelism"] == "dp2"
    assert line["n_gpus"] == 1 and "gloo grad all-reduce" in cfg["workload"]
    assert line["value"] > 0 and line["steps"] == 16 and line["scaling"] == "weak"


def test_bench_strong_scaling_shards_the_job_over_the_ranks():
    """`--scaling strong`: --num-envs / --replay / --batch are the JOB's sizes (BASELINE configs[3] as written) and every rank takes
    1 / N of each; `value` then counts steps of the job's batch, not batch-sized steps summed over ranks."""
    line = _run_bench("--gpus", "2", "--backend", "gloo", "--share-gpu", "--scaling", "strong")
    cfg = line["config"]
    assert line["scaling"] == "strong" and cfg["ranks"] == 2
    assert cfg["per_rank"] == {"num_envs": 128, "replay_rows": 10000, "batch": 512}
    assert cfg["job"] == {"num_envs": 256, "replay_rows": 20000, "batch": 1024}
    assert abs(line["value"] - line["steps"] / (line["ms_per_step"] * line["steps"] * 1e-3)) < 1e-6 * line["value"]   # no x ranks
    assert abs(line["env_steps_per_s"] - line["value"] / 8 * 256) < 1e-6 * line["env_steps_per_s"]


def test_bench_split2_layout_rehearsed_on_one_card():
    """BASELINE configs[2] entry: simulator on GPU 0, learners on GPU 1, copy-stream hand-offs.  With --share-gpu both are
    cuda:0 and every hand-off still goes through the Shipper path."""
    line = _run_bench("--gpus", "2", "--layout", "split2", "--share-gpu")
    cfg = line["config"]
    assert cfg["layout"] == "split2" and cfg["ranks"] == 1 and cfg["share_gpu"] is True and line["n_gpus"] == 1
    assert "functional split" in cfg["workload"] and line["value"] > 0
    assert line["roofline"]["frac"] > 0 and line["roofline_gather"]["frac"] > 0


# --------------------------------------------------------------------------- a24: the composed rollout
class _RecordingEnv:
    """The env is an INPUT of the rollout for both sides: record what the product's env returned so the oracle can be
    driven by the very same transitions (the synthetic env's Box-Muller differs in the last bit between GPU and CPU libm)."""

    def __init__(self, env):
        self.env, self.log, self.first_obs = env, [], None
        self.observation_space, self.action_space = env.observation_space, env.action_space
        self.max_episode_length, self.num_envs = env.max_episode_length, env.num_envs

    def reset(self):
        self.first_obs = self.env.reset()
        return self.first_obs

    def step(self, action):
        out = self.env.step(action)
        self.log.append((action.clone(), out[0].clone(), out[1].clone(), out[2].clone()))
        return out


def test_composed_rollout_explore_env_vs_oracle():
    """SURVEY 8(a) row a24, pql/algo/pql_actor.py:87-147 as ONE composition: per env step RunningMeanStd.update(obs) ->
    un-clamped normalise -> policy -> mixed per-env noise (or U(-1,1) in the warm-up) -> env.step -> episode trackers ->
    handle_timeout; then reward scale -> n-step assembly -> the (obs, 5-tuple, steps) hand-off.  Warm-up T = 32, then three
    T = 1 calls, with the noise draws injected.  Bars: emitted n-step rows bit-exact, actions 1e-5, statistics 1e-6,
    tracker windows equal to the reference's deque."""
    from collections import deque
    import detdata as dd
    from oracle import pql_ref_cpu as ref
    from pql_amd.algo.pql_actor import PQLActor
    from pql_amd.envs.synthetic import create_task_env
    from pql_amd.models.mlp import TanhMLPPolicy
    from pql_amd.utils.cfg import load_cfg
    dev = torch.device("cuda:0")
    N, O, A, n = 96, 8, 2, 3
    cfg = load_cfg(["task=Toy", "task.episode_length=6", f"num_envs={N}", "algo.tracker_len=20", "algo.v_learner_gpu=0",
                    "algo.p_learner_gpu=0", "algo.num_gpus=1", "sim_device=cuda:0", "device=cuda:0", f"algo.nstep={n}"])
    cfg.algo.reward_scale = 0.01
    env = _RecordingEnv(create_task_env(cfg))
    actor = PQLActor(env, cfg)
    ast = dd.mlp_state(O, A, 17)
    pol = TanhMLPPolicy((O,), A).to(dev)
    pol.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in ast.items()})
    actor.set_actor(pol)
    actor.reset_agent()

    # ---- oracle state
    apar = ref.params_from_state(ast)
    rms, ns = ref.RunningMeanStdRef((O,)), ref.NStepRef(O, A, N, n)
    ret_win, len_win = deque([0.0] * 20, maxlen=20), deque([0.0] * 20, maxlen=20)
    cur_ret, cur_len = torch.zeros(N), torch.zeros(N)
    g = torch.Generator().manual_seed(3)
    obs = env.first_obs.cpu()
    cursor = 0

    def oracle_call(T, random, draws):
        nonlocal obs, cursor, cur_ret, cur_len
        sl = [torch.zeros((N, T, O)), torch.zeros((N, T, A)), torch.zeros((N, T, 1)), torch.zeros((N, T, O)), torch.zeros((N, T, 1))]
        for t in range(T):
            rms.update(obs)                                                                    # :98-99
            if random:
                act = draws[t] * 2.0 - 1.0                                                     # :100-102
            else:
                act = ref.mixed_noise_ref(ref.actor_forward_ref(apar, ref.normalize_ref(obs, rms.states(), clamp=False)),
                                          draws[t], float(cfg.algo.noise.std_min), float(cfg.algo.noise.std_max))   # :69-85
            logged_act, nobs, rew, done = (x.cpu() for x in env.log[cursor]); cursor += 1
            torch.testing.assert_close(logged_act, act, rtol=0, atol=1e-5)
            done = done.float()
            cur_ret += rew; cur_len += 1                                                       # :129-135
            fin = done.bool()
            ret_win.extend(cur_ret[fin].tolist()); len_win.extend(cur_len[fin].tolist())
            cur_ret[fin] = 0; cur_len[fin] = 0
            sl[0][:, t] = obs; sl[1][:, t] = logged_act; sl[2][:, t, 0] = rew; sl[3][:, t] = nobs; sl[4][:, t, 0] = done
            obs = nobs
        sl[2] = sl[2] * 0.01                                                                    # :116
        return ns.add(*sl)                                                                      # :118

    def check(T, random):
        draws = [torch.rand((N, A), generator=g) if random else torch.randn((N, A), generator=g) for _ in range(T)]
        p_data, v_data, steps = actor.explore_env(env, T, random=random, draws=[d.to(dev) for d in draws])
        want = oracle_call(T, random, draws)
        torch.cuda.synchronize()
        assert steps == T * N and len(v_data) == 5
        for name, got, exp in zip(("obs", "action", "reward", "next_obs", "done"), v_data, want):
            assert got.shape == exp.shape and got.dtype == torch.float32, name
            assert torch.equal(got.cpu(), exp), name                  # emitted n-step rows: bit-exact (time-major order included)
        assert torch.equal(p_data.cpu(), want[0])
        m, v, eps = actor.obs_rms.get_states()
        torch.testing.assert_close(m.cpu(), rms.mean, rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(v.cpu(), rms.var, rtol=1e-5, atol=1e-6)
        assert eps == rms.eps and abs(actor.obs_rms.count - rms.count) < 1e-6 * rms.count
        assert actor.return_tracker.mean() == pytest.approx(float(np.mean(ret_win)), rel=1e-5, abs=1e-7)
        assert actor.step_tracker.mean() == pytest.approx(float(np.mean(len_win)), rel=1e-6)
        assert torch.equal(actor.obs.cpu(), obs)

    check(32, True)          # warm-up (train_pql.py:57-59): T - n + 1 = 30 emitted blocks of N rows
    for _ in range(3):
        check(1, False)      # steady state: one block of N rows per call
    assert sum(x != 0 for x in len_win) > 10   # episodes did finish (mean length 6): the trackers were exercised


def test_rollout_policy_forward_follows_every_way_its_weights_can_change():
    """The rollout's fused policy forward reads a fragment-ordered COPY of the replica's weights.  It must follow the replica
    however it changes: `set_actor` (fenced adoption), the reference idiom `pql_actor.actor = deepcopy(actor)` (train_pql.py:52,
    109), and an in-place load between two `explore_env` calls; and it must equal the module's own (per-layer) forward."""
    from copy import deepcopy
    import detdata as dd
    from pql_amd.algo.pql_actor import PQLActor
    from pql_amd.envs.synthetic import create_task_env
    from pql_amd.models.mlp import TanhMLPPolicy
    from pql_amd.utils.cfg import load_cfg
    dev = torch.device("cuda:0")
    N, O, A = 256, 88, 16
    cfg = load_cfg(["task=AllegroHand", f"num_envs={N}", "algo.v_learner_gpu=0", "algo.p_learner_gpu=0", "algo.num_gpus=1",
                    "sim_device=cuda:0", "device=cuda:0"])
    actor = PQLActor(create_task_env(cfg), cfg)
    actor.reset_agent()

    def policy(seed):
        m = TanhMLPPolicy((O,), A).to(dev)
        m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in dd.mlp_state(O, A, seed).items()})
        return m

    obs = torch.from_numpy(dd.uniform((N, O), 5, -2, 2)).to(dev)

    def check(module, what):
        got = actor.get_actions(obs, sample=False)
        with torch.no_grad():
            want = module(actor.obs_rms.normalize(obs))
        assert actor._pk is not None and actor._pk.tensor is not None, "fused policy forward not in use"
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=what)

    a, b, c = policy(17), policy(18), policy(19)
    actor.set_actor(a); check(a, "set_actor (first)")
    actor.set_actor(b); check(b, "set_actor (adoption into the replica)")
    actor.actor = deepcopy(c); check(c, "assignment of .actor")
    actor.actor.load_state_dict(a.state_dict())          # in place: picked up by the next explore_env call
    actor.explore_env(actor.env, 4, random=False)         # (the first call must span the n-step window)
    obs = torch.from_numpy(dd.uniform((N, O), 6, -2, 2)).to(dev)
    check(a, "in-place load before explore_env")
    with torch.no_grad():
        assert not torch.allclose(a(obs), b(obs))        # the policies do differ


def test_published_snapshot_reads_wait_for_the_learner_stream():
    """state_dict() / forward / the evaluator's spec of a PUBLISHED snapshot are enqueued on the caller's stream while the
    snapshot is filled on the learner's: with the learner's queue stalled the reads must still see the published weights
    (round-2 advisor finding: only deepcopy and adopt_arena honoured the lease)."""
    from pql_amd.models.mlp import DoubleQ, TanhMLPPolicy
    from pql_amd.utils import handoff as H
    from pql_amd.utils.evaluator import module_to_spec
    dev = torch.device("cuda:0")
    for cls, args in ((DoubleQ, (8, 2)), (TanhMLPPolicy, (8, 2))):
        live = cls(*args, hidden_layers=[64, 32]).to(dev)
        pub = H.ArenaPublisher(live)
        side = torch.cuda.Stream(dev)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            torch.cuda._sleep(400_000_000)          # ~0.2 s: the learner's queue is far behind the host
            live.arena.data.fill_(0.25)             # "an optimiser step"
            snap = pub.publish()
        sd = snap.state_dict()                      # caller's (default) stream
        spec = module_to_spec(snap)
        x = torch.ones((4, 8), device=dev)
        y = snap(x) if cls is TanhMLPPolicy else snap.get_q1(x, torch.ones((4, 2), device=dev))
        with torch.cuda.stream(side):               # the publisher's next refill of that slot must wait for those reads
            live.arena.data.fill_(-1.0)
            pub.publish(); pub.publish()
        torch.cuda.synchronize()
        for k, t in sd.items():
            assert torch.all(t == 0.25), k
        for k, t in spec["state"].items():
            assert torch.all(t == 0.25), k
        ref = cls(*args, hidden_layers=[64, 32]).to(dev)
        ref.arena.data.fill_(0.25)
        want = ref(x) if cls is TanhMLPPolicy else ref.get_q1(x, torch.ones((4, 2), device=dev))
        torch.testing.assert_close(y, want, rtol=0, atol=0)


def _run_train_pql_ranks(world, overrides, port, timeout=600):
    """`python -m torch.distributed.run --nproc-per-node <world> scripts/train_pql.py ...` as a CHILD process (the ranks initialise the
    GPU themselves) -> the per-rank result dicts, ordered by rank."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "scripts", "train_pql.py"), *overrides],
                       env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stderr[-3000:], r.stdout[-1000:])
    import re
    outs = [json.loads(m.group(1)) for m in re.finditer(r"TRAIN_PQL_RESULT (\{.*?\})\s*(?=TRAIN_PQL_RESULT|$)", r.stdout, re.S)]
    assert len(outs) == world, r.stdout[-2000:]
    return sorted(outs, key=lambda o: o["rank"])


@pytest.mark.parametrize("distl", [False, True])
def test_train_pql_data_parallel_branch_two_ranks_on_one_card(distl):
    """scripts/train_pql.py's WORLD_SIZE > 1 branch (train_pql.py: init_data_parallel, shard, component_groups, the rank-0 broadcast of
    the initial arenas, the merged running statistics, rank-0-only evaluator / logging, the stop agreement and the barrier before
    destroy_process_group) executed END TO END with two ranks: strong mode (the job's 128 envs / 8000 rows / batch 512 split in
    two), algo.dp_backend=gloo + algo.dp_share_gpu=True because this box has one card and RCCL refuses two ranks on one device.
    Both ranks must leave the loop in the same iteration with the same counters, and -- replicated optimiser on all-reduced
    gradients -- with BIT-EQUAL critic, target and actor arenas."""
    outs = _run_train_pql_ranks(2, ["task.name=Toy", "num_envs=128", "algo.batch_size=512", "algo.memory_size=8000", "max_step=12000",
                                    "algo.dp_backend=gloo", "algo.dp_share_gpu=True", f"algo.distl={distl}", "algo.graph=True",
                                    "algo.eval_freq=20", "algo.log_freq=10"], port=29611 + int(distl))
    a, b = outs
    assert (a["rank"], b["rank"], a["world"], b["world"]) == (0, 1, 2, 2)
    for k in ("global_steps", "critic_updates", "actor_updates", "rollout_iterations", "critic_sha", "critic_target_sha", "actor_sha"):
        assert a[k] == b[k], (k, a[k], b[k])
    iters = a["rollout_iterations"]
    assert a["critic_updates"] == 8 * iters and a["actor_updates"] == 4 * iters and a["global_steps"] >= 12000
    assert a["global_steps"] == 128 * 32 + iters * 128        # JOB-wide env steps: warm-up (32 steps) + one step of all 128 envs per iteration
    assert np.isfinite(a["critic_loss"]) and np.isfinite(b["actor_loss"])


def test_train_pql_data_parallel_wall_clock_stop_is_agreed():
    """max_step unset: the stop criterion is wall-clock time, which differs between ranks -- rank 0's verdict is broadcast
    (agree_to_stop) so that nobody enters an iteration whose collectives the other rank will not join.  Weak mode here."""
    outs = _run_train_pql_ranks(2, ["task.name=Toy", "num_envs=64", "algo.batch_size=256", "algo.memory_size=4000", "max_time=4",
                                    "algo.dp_backend=gloo", "algo.dp_share_gpu=True", "algo.dp_global=False", "algo.eval_freq=1000000",
                                    "algo.log_freq=1000000"], port=29621)
    a, b = outs
    assert a["rollout_iterations"] == b["rollout_iterations"] > 3 and a["critic_sha"] == b["critic_sha"] and a["actor_sha"] == b["actor_sha"]
    assert a["global_steps"] == 2 * (64 * 32 + a["rollout_iterations"] * 64)   # weak: every rank brings its own 64 envs


def test_train_pql_data_parallel_four_ranks_on_one_card():
    """The same branch with FOUR ranks (more than two: the host-staged ring all-reduce then sums the ranks' addends in an order
    that differs from element to element, and every rank must still receive the same bits): strong mode, 128 envs / 8000 rows /
    batch 512 split in four; equal counters and bit-equal arenas on all four ranks."""
    outs = _run_train_pql_ranks(4, ["task.name=Toy", "num_envs=128", "algo.batch_size=512", "algo.memory_size=8000", "max_step=8000",
                                    "algo.dp_backend=gloo", "algo.dp_share_gpu=True", "algo.graph=True", "algo.eval_freq=1000000",
                                    "algo.log_freq=1000000"], port=29631)
    assert [o["rank"] for o in outs] == [0, 1, 2, 3] and all(o["world"] == 4 for o in outs)
    for k in ("global_steps", "critic_updates", "actor_updates", "rollout_iterations", "critic_sha", "critic_target_sha", "actor_sha"):
        assert len({o[k] for o in outs}) == 1, (k, [o[k] for o in outs])
    iters = outs[0]["rollout_iterations"]
    assert outs[0]["critic_updates"] == 8 * iters and outs[0]["global_steps"] == 128 * 32 + iters * 128
