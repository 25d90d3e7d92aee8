#!/usr/bin/env python3
"""PQL learner throughput on MI355X: learner grad-steps/sec + env-steps/sec at 4096 envs, batch 8192.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One bench "step" = one slice of the PQL schedule at the reference's design ratios (pql_algo.yaml:17-18,
train_pql.py:133,145): ONE V-learner gradient step on a batch of 8192, plus one P-learner gradient step every
2nd step, plus one env iteration (4096 synthetic env steps -> running obs stats -> n-step assembly -> insert
into the V and P replay rings -> weight hand-off) every 8th step.  So
    value            = V-learner grad-steps/s (whole job)
    p_grad_steps/s   = value / 2 ;  env_steps/s = value / 8 * num_envs
Workload = BASELINE.json configs[1]: obs 88, act 16, replay 1M rows pre-filled to capacity and resident in HBM,
DoubleQ MLP [512,512,256], n-step 3, synthetic transitions.

N > 1 (`--layout dp`, default): data-parallel weak scaling.  Every rank owns 4096 envs, a 1M-row replay shard and a
batch of 8192; the only collective on the data path is the RCCL all-reduce of the flat gradient arena before the
(replicated) optimiser step.  `value` then counts batch-8192 gradient steps summed over ranks (N x steps / time).
Started without torchrun (`python bench.py --gpus N`), this script launches the N ranks itself (a torchrun child
process, before anything here touches a GPU).

`--gpus 2 --layout split2`: BASELINE configs[2], the reference's default placement -- simulator + rollout policy on GPU 0,
V-learner and P-learner on GPU 1, parameters and transitions shipped over xGMI by copy streams (one process, same metric).
`--share-gpu` rehearses either layout on ONE card (and says so in `config`; `n_gpus` then stays 1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz x 256 FLOP/clk
PEAK_HBM_GBS = 8000.0

TASKS = {"AllegroHand": (88, 16), "ShadowHand": (211, 20), "Humanoid": (108, 21), "Toy": (8, 2)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=48)
    ap.add_argument("--task", default="AllegroHand")
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--replay", type=int, default=1_000_000)
    ap.add_argument("--nstep", type=int, default=3)
    ap.add_argument("--hidden", default="512,512,256", help="BASELINE shape; the reference default is 512,256,128")
    ap.add_argument("--distl", action="store_true")
    ap.add_argument("--burn-in-ms", type=float, default=120.0,
                    help="set-up: keep the matrix pipes busy this long (the V step's MFMA launches on scratch buffers; no learner "
                         "state is touched) right before the warm-up steps.  After ANY idle gap of >= 10 ms the MFMA kernels of this "
                         "GPU run ~14 %% slow and recover over the next ~30 ms of load (measured in round 2), and set-up "
                         "ends in such gaps: without this a 20-step timed region sits on that ramp.  Recorded in config.burn_in_ms; "
                         "0 disables")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--rng", default="auto", choices=["auto", "torch", "philox"],
                    help="auto: the learners' draws of the next 8 (V) / 4 (P) steps from one libpqlk launch (torch's own numbers, verified on "
                         "the device) + one batched replay gather; torch: randint + normal_ ATen launches and a gather per step (round 2)")
    ap.add_argument("--graph-rng", action="store_true", help="capture the learners' RNG draws inside their hipGraphs (A/B; default: in front)")
    ap.add_argument("--no-streams", action="store_true", help="serialise V / P / rollout on one stream")
    ap.add_argument("--no-fused", action="store_true", help="per-layer GEMM launches instead of the fused hidden-layer forward")
    ap.add_argument("--no-fused-tail", action="store_true", help="separate loss-fold / gradient-norm launches (A/B of algo.fused_tail)")
    ap.add_argument("--dp-buckets", default="auto", choices=["auto", "layer", "one"],
                    help="data parallel: the critic's gradient all-reduce in per-layer buckets issued behind each layer's slab sum, "
                         "as one collective after the whole backward, or auto = layer iff PQL_DP_GRAPH_COLLECTIVE=1 (algo.dp_buckets)")
    ap.add_argument("--no-run-graph", action="store_true", help="one hipGraph per learner step instead of one per run of steps between two hand-offs (A/B of algo.run_graph)")
    ap.add_argument("--no-td-forward", action="store_true", help="head backward as its own launch instead of inside the critic's forward (A/B of algo.td_in_forward)")
    ap.add_argument("--override", action="append", default=[], metavar="KEY=VALUE",
                    help="extra cfg override in the entry point's syntax (A/B switches: algo.actor_ahead=False, algo.dpg_fused=False ...); "
                         "recorded in config.overrides")
    ap.add_argument("--gather-event-probe", action="store_true", help="(with --no-roofline, under rocprofv3 --kernel-trace) time the K-batch gather "
                    "with HIP events around single eager launches and print it: calibration of roofline_gather's method against the trace")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true",
                    help="skip the roofline / free-running sections after the timed blocks (rocprofv3 kernel-trace runs: the CSV's "
                         "call counts then equal steps x launches per step; use with --burn-in-ms 0)")
    ap.add_argument("--cpu-steps", type=int, default=96, help="schedule steps of the bounded CPU-oracle sample (~15 s)")
    ap.add_argument("--v-only", action="store_true", help="time free-running V-learner steps only")
    ap.add_argument("--p-only", action="store_true", help="time free-running P-learner steps only (profiling)")
    ap.add_argument("--layout", default="dp", choices=["dp", "split2"],
                    help="N>1: dp = one rank per GPU, env/replay shards + RCCL gradient all-reduce; split2 = simulator on GPU 0, "
                         "both learners on GPU 1 (BASELINE configs[2])")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N>1 data parallel: weak = --num-envs / --replay / --batch are PER RANK (the job grows with N; `value` counts "
                         "batch-sized steps summed over ranks); strong = they are the JOB's sizes and every rank takes 1/N of each "
                         "(BASELINE configs[3] as written: --num-envs 16384 --gpus 8 = 2048 envs per rank)")
    ap.add_argument("--repeat", type=int, default=5, help="timed blocks of --steps steps; `value` is the FIRST block, "
                                                          "median/min/max over all blocks are reported beside it")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); 'gloo' + --share-gpu rehearses the DP path on one GPU")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    return ap.parse_args()


def mlp_macs(dims):
    return sum(a * b for a, b in zip(dims[:-1], dims[1:]))


def flops_per_step(O, A, hidden, out_c, B):
    """fwd = 2*MAC, bwd = 2 x fwd.  V-step = 2B(MAC_actor + 4*MAC_critic2); P = 2B(3 MAC_actor + 2 MAC_critic2)."""
    mac_a = mlp_macs([O, *hidden, A])
    mac_c2 = 2 * mlp_macs([O + A, *hidden, out_c])
    return 2 * B * (mac_a + 4 * mac_c2), 2 * B * (3 * mac_a + 2 * mac_c2)


def build_system(args, rank, world, device, pg, learner_device=None):
    from pql_amd.algo.pql_actor import PQLActor
    from pql_amd.algo.pql_p_learner import PQLPLearner
    from pql_amd.algo.pql_v_learner import PQLVLearner
    from pql_amd.envs.synthetic import create_task_env
    from pql_amd.utils.cfg import load_cfg
    hidden = [int(x) for x in args.hidden.split(",")]
    ov = [f"num_envs={args.num_envs}", f"task.name={args.task}", f"algo.batch_size={args.batch}",
          f"algo.memory_size={args.replay}", f"algo.nstep={args.nstep}", f"algo.distl={args.distl}",
          f"algo.v_learner_gpu={(learner_device or device).index}", f"algo.p_learner_gpu={(learner_device or device).index}",
          f"algo.num_gpus={1 if learner_device is None else 2}",
          f"algo.graph={not args.no_graph}", f"algo.graph_rng={bool(getattr(args, 'graph_rng', False))}", f"algo.streams={not args.no_streams}", f"algo.fused={not args.no_fused}", f"sim_device=cuda:{device.index}",
          f"device=cuda:{device.index}", *getattr(args, "override", [])]
    cfg = load_cfg(ov)
    cfg.algo.hidden_layers = hidden
    cfg.algo.fused_tail = not getattr(args, "no_fused_tail", False)
    cfg.algo.dp_buckets = getattr(args, "dp_buckets", "auto")
    cfg.algo.td_in_forward = not getattr(args, "no_td_forward", False)
    cfg.algo.run_graph = not getattr(args, "no_run_graph", False)
    cfg.algo.rng = getattr(args, "rng", "auto")
    cfg.algo.reward_scale = 0.01   # preprocess_cfg's AllegroHand value (common.py:159-170)
    sh = getattr(args, "shard", None)
    env_offset = sh.env_offset if sh is not None else rank * args.num_envs
    total_envs = sh.total_envs if sh is not None else world * args.num_envs
    env = create_task_env(cfg, env_offset=env_offset)
    actor = PQLActor(env, cfg, env_offset=env_offset, total_envs=total_envs)
    # one communicator per collective-issuing component, so the three queues do not serialise on one internal RCCL stream
    from pql_amd.utils.dp import component_groups
    groups = component_groups(pg)
    v = PQLVLearner(env.observation_space.shape, env.action_space.shape[0], cfg, process_group=groups["v"])
    p = PQLPLearner(env.observation_space.shape, env.action_space.shape[0], cfg, process_group=groups["p"])
    if pg is not None:   # replicated parameters: every rank starts from rank 0's weights
        from pql_amd.utils.dp import broadcast_from_rank0
        for t in (v.critic.arena.data, p.actor.arena.data):
            broadcast_from_rank0(t, pg)
        v.critic_target.arena.data.copy_(v.critic.arena.data)
        actor.obs_rms.pg = groups["rms"]
    return cfg, env, actor, v, p


def prefill(actor, v, p, env, cfg, args, device, prepare=("v", "p")):
    """Warm-up rollout (train_pql.py:57-68), then fill both replay rings to capacity with synthetic rows so the
    randint bound is constant and samples come from HBM, not cache.  Returns the newest (critic, policy) snapshots."""
    critic, _, _ = v.start()
    pol, _, _ = p.start()
    actor.set_actor(pol)
    actor.reset_agent()
    p_data, v_data, _ = actor.explore_env(env, cfg.algo.warm_up, random=True)
    critic, _, _ = v.update(pol, v_data, actor.obs_rms.get_states(v.device), 0)
    pol, _, _ = p.update(critic, p_data, actor.obs_rms.get_states(p.device), 0)
    O, A = env.obs_dim, env.act_dim
    chunk = 65536
    ldev = v.device
    g = torch.Generator(device=ldev)
    g.manual_seed(1234 + ldev.index)
    while not v.memory.if_full:
        m = min(chunk, v.memory.capacity)
        obs = torch.randn((m, O), device=ldev, generator=g)
        traj = (obs, torch.rand((m, A), device=ldev, generator=g) * 2 - 1, torch.randn((m, 1), device=ldev, generator=g) * 0.01,
                torch.randn((m, O), device=ldev, generator=g), (torch.rand((m, 1), device=ldev, generator=g) < 1 / 300).float())
        critic, _, _ = v.update(pol, traj, actor.obs_rms.get_states(v.device), 0)
        pol, _, _ = p.update(critic, obs, actor.obs_rms.get_states(p.device), 0)
    # set-up, like the ring fill: a few rollout iterations at the schedule's own horizon with their hand-offs, so that the
    # trajectory slabs, the three round-robin output blocks, the statistics buffers and the allocator's cache exist before the
    # first warm-up step (each first-time allocation is a device-synchronising hipMalloc; measured: without this the first
    # ~40 schedule steps run 5-15 % slow), and the learners' workspaces and hipGraphs (capture restores everything it touches)
    for _ in range(4):
        actor.set_actor(pol)
        p_data, v_data, _ = actor.explore_env(env, cfg.algo.horizon_len, random=False)
        critic, _, _ = v.update(pol, v_data, actor.obs_rms.get_states(v.device), 0)
        pol, _, _ = p.update(critic, p_data, actor.obs_rms.get_states(p.device), 0)
    if "v" in prepare:
        v.prepare()
    if "p" in prepare:   # (a --v-only profile run never steps the P-learner, not even inside its graph capture)
        p.prepare()
    sync_all(device, ldev)
    return critic, pol


def sync_all(*devices):
    for d in {torch.device(x) for x in devices}:
        torch.cuda.synchronize(d)


class Schedule:
    """The design schedule: 1 env iteration : 4 P-steps : 8 V-steps (critic_sample_ratio 8, critic_actor_ratio 2).  Hand-offs go through the learners' `update()` exactly as in
    scripts/train_pql.py: event-fenced, double-buffered, snapshots of the weights (pql_amd/utils/handoff.py)."""

    def __init__(self, actor, v, p, env, cfg, device, critic, policy, mode="schedule"):
        self.actor, self.v, self.p, self.env, self.cfg, self.device, self.mode = actor, v, p, env, cfg, device, mode
        self.k = 0
        self.r_p = int(cfg.algo.critic_actor_ratio)
        self.r_env = int(cfg.algo.critic_sample_ratio)
        self.global_steps = 0
        self.pending = None
        self.critic, self.policy = critic, policy   # newest snapshots handed out by the learners

    def align(self):
        """Skip to the next slice boundary without issuing anything (the steps of the current slice were issued, and counted
        by nobody, when it began): what follows is timed from a boundary."""
        rr = self.r_env if self.mode != "p_only" else max(self.r_env // self.r_p, 1)
        self.k = (self.k + rr - 1) // rr * rr

    def step(self, remaining=None):
        """One V-learner step of the schedule.  Rollout is software-pipelined one slice ahead, like the reference's asynchronous
        actor: the transitions handed to the learners at slice i were produced (on the rollout queue) while the learners ran
        slice i-1.  The learner steps of a slice are ISSUED together at its first step (`learn_many`: one hipGraph per learner and
        slice instead of one per step; same launches in the same order on each learner's queue), the other steps of the slice
        issue nothing; `remaining` = steps the caller will still make (this one included), so that a block which ends inside a
        slice issues exactly its own steps."""
        k = self.k
        self.k += 1
        r = self.r_env
        if remaining is None:
            remaining = r
        if self.mode in ("v_only", "p_only"):
            rr = r if self.mode == "v_only" else max(r // self.r_p, 1)
            if k % rr == 0:
                (self.v if self.mode == "v_only" else self.p).learn_many(min(rr, remaining))
            return
        if k % self.r_env == 0:
            # Issue order: each learner's hand-off is followed at once by its steps, the rollout of the NEXT slice goes last.  The
            # data flow is the one of scripts/train_pql.py (the snapshots `update()` returns are copies taken at the hand-off, so
            # it does not matter to P or to the rollout that V's steps are already queued behind it); what changes is that
            # after the barrier + synchronise that opens a timed block the learners' queues do not sit idle while the host is
            # still enqueuing the ~20 launches of the rollout (0.3 ms of a 17-ms block at --steps 20; nothing in steady state,
            # where the host runs a slice ahead of the device).
            nv = min(r, remaining)
            rms = self.actor.obs_rms
            if self.pending is not None:
                p_data, v_data = self.pending
                self.critic, _, _ = self.v.update(self.policy, v_data, rms.get_states(self.v.device), 0)   # transitions + policy -> V
            self.v.learn_many(nv)
            if self.pending is not None:
                self.policy, _, _ = self.p.update(self.critic, p_data, rms.get_states(self.p.device), 0)   # obs + critic -> P
            self.p.learn_many(nv // self.r_p)
            self.actor.set_actor(self.policy)                                                               # policy -> rollout
            p_data, v_data, n = self.actor.explore_env(self.env, self.cfg.algo.horizon_len, random=False)
            self.global_steps += n
            self.pending = (p_data, v_data)


def v_section(v):
    """The forward + backward launches of ONE V-learner step as a closure (the learner's own calls, on scratch state: no optimiser
    step, no parameter changes): actor forward, target twin forward, twin forward (with the Q head's backward inside), the five
    backward GEMMs and the slab sum that closes the backward call."""
    from pql_amd import _lib as L
    import ctypes as C
    ws = v._workspace(int(v.cfg.algo.batch_size))
    B = ws["B"]
    al, cl = v.actor.layout, v.critic.layout
    st = lambda: L.stream(v.device)  # noqa: E731
    O = v.memory.ring.O

    def section():
        from pql_amd.models.mlp import mlp_forward_raw
        fused_actor = v.pk_actor is not None and v.pk_actor.tensor is not None
        if not ws.get("actor_ahead"):   # (algo.actor_ahead: the K steps' target actions come from one launch at prefetch time: actor_ahead_ms)
            mlp_forward_raw(al, v.actor.arena.data, ws["xn_sa"] if fused_actor else ws["xn_obs"], L.ACT_TANH_NOISE, ws["draw"], 0.8, 0.2,
                            ws["acts_a"], ws["xn_sa"][:, O:], packed=v.pk_actor, stash_all=False)
        mlp_forward_raw(cl, v.critic_target.arena.data, ws["xn_sa"], L.ACT_NONE, acts=ws["acts_t"], packed=v.pk_target, stash_all=False)
        if ws.get("td_fwd", 0) > 0:   # the learner's own pair of calls: the Q head's backward rides in the critic's forward launch
            L.check(L.lib.pqlk_mlp_forward_td(C.byref(cl.desc), L.ptr(v.critic.arena.data), L.ptr(v.pk_critic.tensor), L.ptr(ws["x_sa"]), ws["ld_sa"],
                                              B, L.ptr(ws["acts_c"]), L.ptr(ws["acts_t"]), L.ptr(ws["rew"]), L.ptr(ws["done"]), 0.97,
                                              L.ptr(ws["scratch"]), L.ptr(ws["bwd"]), ws["bwd"].numel(), ws["splits"], st()))
            L.check(L.lib.pqlk_mlp_backward_td_tail(C.byref(cl.desc), L.ptr(v.critic.arena.data), L.ptr(ws["x_sa"]), ws["ld_sa"], B,
                                                    L.ptr(ws["acts_c"]), L.ptr(ws["grads"]), ws["splits"], L.ptr(ws["bwd"]), ws["bwd"].numel(),
                                                    None, None, st()))
            return
        mlp_forward_raw(cl, v.critic.arena.data, ws["x_sa"], L.ACT_NONE, acts=ws["acts_c"], packed=v.pk_critic, stash_all=True)
        L.check(L.lib.pqlk_mlp_backward(C.byref(cl.desc), L.ptr(v.critic.arena.data), L.ptr(ws["x_sa"]), ws["ld_sa"], B,
                                        L.ptr(ws["acts_c"]), L.ptr(ws["dy"]), L.ptr(ws["grads"]), ws["splits"], None, 0, 0, 0,
                                        None, 0, L.ptr(ws["bwd"]), ws["bwd"].numel(), st()))
    return section


def actor_ahead_ms(v, iters=8):
    """Device time of the ONE target-policy forward that serves the next K V-learner steps (PQLVLearner._prefetch, algo.actor_ahead):
    K x B rows, tanh + target-policy noise, actions dropped into the K target-critic input tiles.  HIP events around a hipGraph
    of `iters` launches."""
    from pql_amd import _lib as L
    from pql_amd.models.mlp import mlp_forward_raw
    ws = v._workspace(int(v.cfg.algo.batch_size))
    K, B, O = ws["K"], ws["B"], v.memory.ring.O
    xn = ws["xn_sa_all"].view(K * B, ws["ld_sa"])
    draw = v._ahead.normal.view(K * B, -1)

    def one():
        mlp_forward_raw(v.actor.layout, v.actor.arena.data, xn, L.ACT_TANH_NOISE, draw, 0.8, 0.2, ws["a_out_all"], xn[:, O:], packed=v.pk_actor,
                        stash_all=2)
    with torch.cuda.device(v.device):
        g = _graph_of(lambda: [one() for _ in range(iters)], v.device)
        return _replay_ms(g) / iters


def gemm_section_ms(v, iters=20):
    """Device time of the forward + backward launches of ONE V-learner step (v_section: the slab sum that closes the backward call is
    the one non-MFMA launch in the interval, ~11 us: the fraction is that much conservative) -- measured with HIP events on the stream
    they are launched on (torch's current stream).  With algo.actor_ahead the target policy's forward is one launch per K steps: 1 / K
    of its time is added (actor_ahead_ms)."""
    section = v_section(v)
    ahead = actor_ahead_ms(v) / v._workspace(int(v.cfg.algo.batch_size))["K"] if v._workspace(int(v.cfg.algo.batch_size)).get("actor_ahead") else 0.0

    with torch.cuda.device(v.device):
        for _ in range(3):
            section()
        # replayed from a hipGraph, like the learner's own step: the interval then holds device time only (launched eagerly,
        # a host hiccup between two of the 140 ctypes calls shows up as GPU idle time inside the HIP-event interval)
        side = torch.cuda.Stream(v.device)
        side.wait_stream(torch.cuda.current_stream(v.device))
        with torch.cuda.stream(side):
            section()
        torch.cuda.current_stream(v.device).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):   # (a process-group watchdog thread may poll events meanwhile)
            for _ in range(iters):
                section()
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / iters + ahead


def dominant_kernel_ms(v, iters=20):
    """Device time of ONE launch of the dominant kernel -- the fused twin-critic forward `k_mlp_fwd_fused<2,2>` (two per V
    step, ~38 % of the step) -- measured with HIP events on the stream it is launched on, from a hipGraph of `iters`
    launches so that host call overhead does not leak into the interval."""
    from pql_amd import _lib as L
    from pql_amd.models.mlp import mlp_forward_raw
    ws = v._workspace(int(v.cfg.algo.batch_size))
    cl = v.critic.layout

    def one():
        mlp_forward_raw(cl, v.critic_target.arena.data, ws["xn_sa"], L.ACT_NONE, acts=ws["acts_t"], packed=v.pk_target, stash_all=False)

    with torch.cuda.device(v.device):
        for _ in range(3):
            one()
        side = torch.cuda.Stream(v.device)
        side.wait_stream(torch.cuda.current_stream(v.device))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            one()
        torch.cuda.current_stream(v.device).wait_stream(side)
        with torch.cuda.graph(g, capture_error_mode="thread_local"):   # (a process-group watchdog thread may poll events meanwhile)
            for _ in range(iters):
                one()
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / iters


def gather_ms(v, batches=1, iters=50):
    """Device time of one fused replay gather launch of `batches` x B rows (HIP events, same stream): batches = 1 is the
    per-step launch, batches = K the launch that serves the next K V-learner steps at once (PQLVLearner._prefetch)."""
    from pql_amd import _lib as L
    from pql_amd.algo.pql_v_learner import GATHER_FLAGS
    import ctypes as C
    ws = v._workspace(int(v.cfg.algo.batch_size))
    B = ws["B"]
    rows = batches * B
    mean, var, eps = v._norm_ptrs()
    iters = max(4, min(iters, (1 << 24) // rows))
    idx = torch.randint(v.memory.cur_capacity, size=(iters + 3, rows), device=v.device)
    fused_actor = v.pk_actor is not None and v.pk_actor.tensor is not None
    f = dict(dtype=torch.float32, device=v.device)
    if batches <= ws["K"]:
        x_sa, xn_sa, rew, done = ws["x_sa_all"], ws["xn_sa_all"], ws["rew_all"], ws["done_all"]
    else:
        x_sa, xn_sa = torch.zeros((rows, ws["ld_sa"]), **f), torch.zeros((rows, ws["ld_sa"]), **f)
        rew, done = torch.zeros(rows, **f), torch.zeros(rows, **f)
    xn_obs = None if fused_actor else torch.zeros((rows, ws["ld_o"]), **f)

    def one(i):
        L.check(L.lib.pqlk_replay_gather_fused(C.byref(v.memory.ring.desc), L.ptr(idx[i]), rows, L.ptr(mean), L.ptr(var), eps, GATHER_FLAGS,
                                               L.ptr(x_sa), ws["ld_sa"], L.ptr(xn_sa), L.ptr(xn_obs), ws["ld_o"],
                                               L.ptr(rew), L.ptr(done), L.stream(v.device)))
    with torch.cuda.device(v.device):
        for i in range(3):
            one(i)
        # the launches are replayed from a hipGraph so the host's ctypes call overhead (~10 us, comparable to the kernel)
        # does not leak into the HIP-event interval
        side = torch.cuda.Stream(v.device)
        side.wait_stream(torch.cuda.current_stream(v.device))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            one(0)
        torch.cuda.current_stream(v.device).wait_stream(side)
        with torch.cuda.graph(g, capture_error_mode="thread_local"):   # (a process-group watchdog thread may poll events meanwhile)
            for i in range(iters):
                one(3 + i)
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / iters


def _graph_of(fn, device):
    """hipGraph of `fn()` (warm-up on a side stream first, as torch requires)."""
    side = torch.cuda.Stream(device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream(device).wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        fn()
    return g


def _replay_ms(g, reps=5):
    """Median HIP-event time of `reps` replays of a graph."""
    g.replay()
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        out.append(e0.elapsed_time(e1))
    return sorted(out)[len(out) // 2]


def isolated_launch_ms(launch, device, n=16, sets=4, context=None, evict_mb=384):
    """Device time of ONE launch the way the schedule pays for it: alone, with other work between two launches of the kernel.
    `launch(i, s)` issues the launch with index row i into output-tile set s; `context()` issues the work that separates two launches
    -- the learner step's own kernels (the gather of the schedule runs once per K steps, behind a step's backward and optimiser
    launches, with the caches full of THEIR data) -- default: a read sweep over `evict_mb` MB (more than L2 + the 256-MB Infinity
    Cache; reads, so the caches are left full of CLEAN lines: behind a 384-MB WRITE sweep the same gather reads 40 us, because
    every line it allocates first pushes a dirty line out -- tools/probes/gather_limit_probe.hip).  Two hipGraphs are timed with HIP
    events on the stream they run on -- n x [context, launch] and n x [context] -- and the difference / n is the launch.  Replayed
    back to back (50 launches into one set of output tiles, which then lives in the Infinity Cache) the same kernel reads ~15 %
    faster: that figure is reported as `frac_back_to_back`.  Consecutive launches write `sets` distinct output-tile sets."""
    if context is None:
        evict = torch.zeros(evict_mb * (1 << 20) // 4, dtype=torch.float32, device=device)
        sink = torch.zeros((), dtype=torch.float32, device=device)

        def context():
            torch.sum(evict, dim=0, out=sink)

    def with_launch():
        for i in range(n):
            context()
            launch(i, i % sets)

    def without():
        for _ in range(n):
            context()

    with torch.cuda.device(device):
        ga, gb = _graph_of(with_launch, device), _graph_of(without, device)
        # interleaved replays: clock / thermal drift hits both graphs alike
        ta, tb = [], []
        for _ in range(5):
            ta.append(_replay_ms(ga, 1))
            tb.append(_replay_ms(gb, 1))
        ta.sort(); tb.sort()
    return (ta[2] - tb[2]) / n


def launch_event_us(launch, device, n=16, sets=4, context=None):
    """Duration of ONE launch from HIP events recorded right around it on the stream it runs on, eager (no graph), `context()` between
    two launches as in isolated_launch_ms -- minus the same interval around a one-block no-op launch of this library (what an
    event pair and a dispatch cost by themselves), plus that no-op's own ~1 us.  This is the per-kernel duration rocprofv3
    --kernel-trace reports (begin of the dispatch to its completion signal), obtained in-process."""
    from pql_amd import _lib as L
    tiny_a, tiny_b = torch.zeros(64, device=device), torch.zeros(64, device=device)

    def null():
        L.check(L.lib.pqlk_polyak(L.ptr(tiny_a), L.ptr(tiny_b), 64, 0.5, L.stream(device)))

    def run(fn):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for i, (e0, e1) in enumerate(ev):
            if context is not None:
                context()
            e0.record()
            fn(i)
            e1.record()
        ev[-1][1].synchronize()
        t = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
        return t[len(t) // 2]
    with torch.cuda.device(device):
        run(lambda i: launch(i, i % sets))   # warm
        t_launch = run(lambda i: launch(i, i % sets))
        t_null = run(lambda i: null())
    return t_launch, t_null


NULL_LAUNCH_US = 3.4   # duration rocprofv3 --kernel-trace reports for the one-block no-op below (k_polyak over 64 floats; profiles/README.md)


def learner_gather_us(learner, runs=12):
    """Duration of the replay-gather launch THE LEARNER ITSELF issues -- the K-batch launch of `_prefetch`, into the learner's own
    tiles, between the last step of one run and the first of the next -- while the learner runs its steps the way the schedule
    issues them (`learn_many(K)`): HIP events are recorded right around that one launch on the learner's stream, and right around a
    one-block no-op launch of this library behind it; kernel duration = (interval around the gather) - (interval around the no-op:
    what an event pair + a dispatch cost) + the no-op's own duration (NULL_LAUNCH_US).  Checked against rocprofv3's per-dispatch
    begin/end stamps of the same launches: 27.6 vs 27.3 us (gpurun_out r4j; tests/test_profiles_cpu.py holds the committed bench
    line to the committed kernel trace within 5 %).  Returns (us per gather launch, rows per launch)."""
    from pql_amd import _lib as L
    B = int(learner.cfg.algo.batch_size)
    ws = learner._workspace(B)
    K = int(ws["K"])
    dev = learner.device
    tiny_a, tiny_b = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    orig = learner._gather
    ev, rows = [], [0]

    def timed(ws_, idx, n_rows, *tiles):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record()
        orig(ws_, idx, n_rows, *tiles)
        e[1].record()
        e[2].record()
        L.check(L.lib.pqlk_polyak(L.ptr(tiny_a), L.ptr(tiny_b), 64, 0.5, L.stream(dev)))
        e[3].record()
        ev.append(e)
        rows[0] = int(n_rows)

    learner._gather = timed
    try:
        with torch.cuda.device(dev):
            for _ in range(runs + 2):
                learner.learn_many(K)
            learner.synchronize()
    finally:
        del learner._gather
    tg = sorted(e[0].elapsed_time(e[1]) * 1e3 for e in ev[2:])
    tn = sorted(e[2].elapsed_time(e[3]) * 1e3 for e in ev[2:])
    return tg[len(tg) // 2] - tn[len(tn) // 2] + NULL_LAUNCH_US, rows[0]


def gather_isolated_ms(v, batches, n=16, sets=4, in_step=True):
    """The V-learner's K-batch replay gather as ONE isolated launch (see isolated_launch_ms): between two launches, the forward +
    backward launches of a V step (`in_step`, what the schedule does) or the clean read sweep."""
    from pql_amd import _lib as L
    from pql_amd.algo.pql_v_learner import GATHER_FLAGS
    import ctypes as C
    ws = v._workspace(int(v.cfg.algo.batch_size))
    rows = batches * ws["B"]
    mean, var, eps = v._norm_ptrs()
    idx = torch.randint(v.memory.cur_capacity, size=(n, rows), device=v.device)
    fused_actor = v.pk_actor is not None and v.pk_actor.tensor is not None
    f = dict(dtype=torch.float32, device=v.device)
    tiles = [dict(x_sa=torch.zeros((rows, ws["ld_sa"]), **f), xn_sa=torch.zeros((rows, ws["ld_sa"]), **f), rew=torch.zeros(rows, **f),
                  done=torch.zeros(rows, **f), xn_obs=None if fused_actor else torch.zeros((rows, ws["ld_o"]), **f)) for _ in range(sets)]

    def launch(i, s):
        t = tiles[s]
        L.check(L.lib.pqlk_replay_gather_fused(C.byref(v.memory.ring.desc), L.ptr(idx[i]), rows, L.ptr(mean), L.ptr(var), eps, GATHER_FLAGS,
                                               L.ptr(t["x_sa"]), ws["ld_sa"], L.ptr(t["xn_sa"]), L.ptr(t["xn_obs"]), ws["ld_o"],
                                               L.ptr(t["rew"]), L.ptr(t["done"]), L.stream(v.device)))
    if in_step == "events":
        return launch_event_us(launch, v.device, n, sets, context=v_section(v))
    return isolated_launch_ms(launch, v.device, n, sets, context=v_section(v) if in_step else None)


def gather_p_ms(p, isolated=True, n=16, sets=4):
    """The P-learner's obs gather (K_p batches per launch): isolated launch, or `n` launches back to back."""
    ws = p._workspace(int(p.cfg.algo.batch_size))
    K, B = ws["K"], ws["B"]
    rows = K * B
    idx = torch.randint(p.cur_capacity, size=(n, rows), device=p.device)
    f = dict(dtype=torch.float32, device=p.device)
    tiles = [dict(x_sa=None if ws.get("split_in") else torch.zeros((rows, ws["ld_sa"]), **f), x_obs=torch.zeros((rows, ws["ld_o"]), **f))
             for _ in range(sets)]   # (split input: the P-learner's gather writes the observations once, into the actor's tile)

    def launch(i, s):
        p._gather(ws, idx[i], rows, tiles[s]["x_sa"], tiles[s]["x_obs"])
    if isolated:   # between two launches: a P step's forward + backward launches (scratch state: no optimiser step)
        return isolated_launch_ms(launch, p.device, n, sets, context=lambda: p._step_kernels(ws, None, True, tiles=ws["slots"][0])), rows
    with torch.cuda.device(p.device):
        g = _graph_of(lambda: [launch(i, 0) for i in range(n)], p.device)
        return _replay_ms(g) / n, rows


def free_running(actor, v, p, env, cfg, device, n=160):
    """SURVEY 8(d) asks for the free-running rates as well (the reference with every sleep time forced to 0): each
    component alone on the GPU, back to back, outside the timed region of `value`."""
    out = {}

    def rate(fn, reps, per_call=1.0):
        for _ in range(4):
            fn()
        vals = []
        for _ in range(3):   # median of three blocks: one host hiccup inside a 100-ms block reads as -10 %
            sync_all(device, v.device, p.device)
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            sync_all(device, v.device, p.device)
            vals.append(per_call * reps / (time.perf_counter() - t0))
        return sorted(vals)[1]

    # the learners' steps issued the way the default (fixed-ratio) loop issues them -- one run of critic_sample_ratio steps
    # (critic_sample_ratio / critic_actor_ratio for P) per call, one hipGraph per run -- and one `learn()` call per step, which is
    # what the free-running threads of algo.async_learners=True make
    kv = int(cfg.algo.critic_sample_ratio)
    kp = max(kv // int(cfg.algo.critic_actor_ratio), 1)
    out["v_grad_steps_per_s"] = rate(lambda: v.learn_many(kv), max(n // kv, 4), per_call=float(kv))
    out["p_grad_steps_per_s"] = rate(lambda: p.learn_many(kp), max(n // kp, 4), per_call=float(kp))
    out["v_grad_steps_per_s_one_call_per_step"] = rate(v.learn, n)
    out["p_grad_steps_per_s_one_call_per_step"] = rate(p.learn, n)
    out["env_steps_per_s"] = rate(lambda: actor.explore_env(env, int(cfg.algo.horizon_len), random=False), max(n // 4, 8),
                                  per_call=float(cfg.num_envs) * int(cfg.algo.horizon_len))
    out["note"] = "each component alone, back to back (no ratio control); not part of `value`"
    return out


def free_running_concurrent(actor, v, p, env, cfg, device, critic, policy, seconds=2.0):
    """SURVEY 8(d) "free-running (sleep times forced 0)": the reference's topology with nobody sleeping -- both learners pump
    `learn()` in their own threads (pql_v_learner.py:136-141) while this thread loops rollout -> update -> update -> set_actor
    (train_pql.py:100-119) as fast as it can, all three CONCURRENTLY on the GPU.  Rates = counter deltas / wall time."""
    import threading
    from pql_amd.algo.pql_p_learner import asyn_p_learner
    from pql_amd.algo.pql_v_learner import asyn_v_learner
    stop = threading.Event()
    threads = [threading.Thread(target=asyn_v_learner, args=(v, cfg, stop, 2), daemon=True),
               threading.Thread(target=asyn_p_learner, args=(p, cfg, stop, 2), daemon=True)]
    v.sleep_time = p.sleep_time = 0
    for t in threads:
        t.start()
    rms = actor.obs_rms

    def iteration():
        nonlocal critic, policy
        actor.set_actor(policy)
        p_data, v_data, n = actor.explore_env(env, int(cfg.algo.horizon_len), random=False)
        critic, _, _ = v.update(policy, v_data, rms.get_states(v.device), 0)
        policy, _, _ = p.update(critic, p_data, rms.get_states(p.device), 0)
        return n

    t_end = time.perf_counter() + 0.3   # let the threads reach steady state
    while time.perf_counter() < t_end:
        iteration()
    sync_all(device, v.device, p.device)
    v0, p0, env_steps, iters = v.update_count, p.update_count, 0, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        env_steps += iteration()
        iters += 1
        # (the rollout is host-bound at ~0.2 ms per iteration and would otherwise run hundreds of iterations ahead of the GPU)
        if iters % 8 == 0:
            torch.cuda.current_stream(device).synchronize()
    stop.set()
    for t in threads:
        t.join()
    sync_all(device, v.device, p.device)
    dt = time.perf_counter() - t0
    return {"v_grad_steps_per_s": (v.update_count - v0) / dt, "p_grad_steps_per_s": (p.update_count - p0) / dt,
            "env_steps_per_s": env_steps / dt, "rollout_iterations_per_s": iters / dt, "seconds": dt,
            "note": "V-learner, P-learner (threads) and rollout + hand-offs (this thread) all free-running at once, no sleeps, no "
                    "ratio control; not part of `value`"}


def cpu_baseline(args, O, A, hidden):
    """The CPU oracle (port of the reference learner, pinned to the reference by tests/golden) on this host's
    cores, same schedule, bounded sample."""
    import numpy as np
    from oracle import pql_ref_cpu as ref
    # the GPU box gives one-GPU jobs a 16-core share; more threads than cores makes MKL crawl
    cores = int(os.environ.get("PQL_CPU_THREADS", min(len(os.sched_getaffinity(0)), 16)))
    torch.set_num_threads(cores)
    B, N, cap = args.batch, args.num_envs, min(args.replay, 5_000_000)   # the GPU run's ring size (cfg #2: 1 M rows = 0.8 GB of host memory per copy)
    g = torch.Generator().manual_seed(0)
    dims_a = ref.layer_dims(O, A, hidden)
    dims_c = ref.layer_dims(O + A, 51 if args.distl else 1, hidden)

    def init(dims):
        out = []
        for fi, fo in dims:
            bnd = 1.0 / np.sqrt(fi)
            out += [(torch.rand((fo, fi), generator=g) * 2 - 1) * bnd, (torch.rand((fo,), generator=g) * 2 - 1) * bnd]
        return out

    hp = ref.HyperRef(batch_size=B, distl=args.distl, nstep=args.nstep)
    apar = init(dims_a)
    v = ref.VLearnerRef(O, A, hp, cap, init(dims_c), init(dims_c))
    p = ref.PLearnerRef(O, A, hp, cap, apar)
    ns = ref.NStepRef(O, A, N, args.nstep)
    rms = ref.RunningMeanStdRef((O,))
    rows = cap
    data = (torch.randn((rows, O), generator=g), torch.rand((rows, A), generator=g) * 2 - 1, torch.randn((rows, 1), generator=g) * .01,
            torch.randn((rows, O), generator=g), (torch.rand((rows, 1), generator=g) < 1 / 300).float())
    rms.update(data[0])
    v.update(apar, data, rms.states())
    p.update(v.q1, v.q2, data[0], rms.states())
    warm = [torch.randn((N, args.nstep, O), generator=g), torch.rand((N, args.nstep, A), generator=g), torch.randn((N, args.nstep, 1), generator=g),
            torch.randn((N, args.nstep, O), generator=g), torch.zeros((N, args.nstep, 1))]
    ns.add(*warm)

    def step(k):
        if k % 8 == 0:
            obs = torch.randn((N, 1, O), generator=g)
            rms.update(obs[:, 0])
            act = ref.mixed_noise_ref(ref.actor_forward_ref([q.detach() for q in p.actor], ref.normalize_ref(obs[:, 0], rms.states(), clamp=False)),
                                      torch.randn((N, A), generator=g), 0.05, 0.8)
            out = ns.add(obs, act.unsqueeze(1), torch.randn((N, 1, 1), generator=g) * 0.01, torch.randn((N, 1, O), generator=g),
                         (torch.rand((N, 1, 1), generator=g) < 1 / 300).float())
            v.update([q.detach() for q in p.actor], out, rms.states())
            p.update(v.q1, v.q2, out[0], rms.states())
        v.learn(generator=g)
        if k % 2 == 1:
            p.learn(generator=g)

    for k in range(4):
        step(k)
    t0 = time.perf_counter()
    for k in range(args.cpu_steps):
        step(k)
    dt = time.perf_counter() - t0
    return {"value": args.cpu_steps / dt, "unit": "V-learner grad-steps/s", "cores": cores, "kind": "port",
            "sample": f"{args.cpu_steps} schedule steps (1 V + 1/2 P + 1/8 env iteration each) of oracle/pql_ref_cpu.py, "
                      f"batch {B}, hidden {list(hidden)}, replay {cap} rows, torch {torch.__version__} CPU, {dt:.1f} s"}


def launch_ranks(args):
    """`python bench.py --gpus N` from a bare shell: start the N ranks as a torchrun CHILD process and leave with its exit
    code.  Runs before this process has made any HIP call (device_count() does not initialise the GPU on this image), so
    no process that has touched a GPU ever execs or forks."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if not args.share_gpu and have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible (use --share-gpu to rehearse on one card)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    raise SystemExit(subprocess.run(cmd).returncode)


def main():
    args = parse()
    split = args.layout == "split2"
    if split and args.gpus != 2:
        raise SystemExit("--layout split2 is the two-GPU placement: use --gpus 2")
    if args.gpus > 1 and not split and args.share_gpu and args.backend == "nccl":
        raise SystemExit("--share-gpu puts every rank on cuda:0, which RCCL refuses (duplicate device): add --backend gloo")
    if args.gpus > 1 and not split and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != (1 if split else args.gpus):
        raise SystemExit(f"--gpus {args.gpus} --layout {args.layout} but WORLD_SIZE={world}")
    have = torch.cuda.device_count()
    if not args.share_gpu and have < args.gpus:
        raise SystemExit(f"--gpus {args.gpus}: only {have} GPU(s) visible (use --share-gpu to rehearse on one card)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    device = torch.device("cuda:0" if (args.share_gpu or split) else f"cuda:{local}")
    learner_device = None
    if split:
        learner_device = torch.device("cuda:0" if args.share_gpu else "cuda:1")
        if args.share_gpu:   # one card: still go through the copy streams and landing blocks of the two-GPU path
            from pql_amd.utils import handoff
            handoff.FORCE_SHIP = True
    torch.cuda.set_device(device)
    pg = None
    # The contract is ONE JSON line on stdout.  RCCL prints a version banner (this image exports NCCL_DEBUG=VERSION) and its
    # warnings straight to file descriptor 1: from here on fd 1 is stderr, and the JSON line goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        os.write(json_fd, (json.dumps(line) + "\n").encode())

    if world > 1 or os.environ.get("PQL_FORCE_DP"):   # PQL_FORCE_DP=1: rehearse the RCCL path with a 1-rank group
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=device)   # RCCL
        else:
            torch.distributed.init_process_group(args.backend)
        pg = torch.distributed.group.WORLD
        if torch.distributed.get_world_size(pg) != world:
            raise SystemExit(f"process group has {torch.distributed.get_world_size(pg)} ranks, expected {world}")
    torch.manual_seed(42 + rank)
    # data-parallel shard of this rank (pql_amd/utils/dp.py); from here on args.num_envs / replay / batch are PER RANK
    from pql_amd.utils.dp import shard
    args.shard = shard(args.num_envs, args.replay, args.batch, world, rank, args.scaling if world > 1 else "weak")
    args.num_envs, args.replay, args.batch = args.shard.num_envs, args.shard.memory_size, args.shard.batch_size
    strong = args.shard.scaling == "strong"

    def note(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:6.1f}s] {msg}", file=sys.stderr, flush=True)

    t_start = time.perf_counter()
    cfg, env, actor, v, p = build_system(args, rank, world, device, pg, learner_device)
    note("system built")
    mode = "v_only" if args.v_only else "p_only" if args.p_only else "schedule"
    critic, policy = prefill(actor, v, p, env, cfg, args, device, prepare={"v_only": ("v",), "p_only": ("p",)}.get(mode, ("v", "p")))
    note(f"replay pre-filled: {v.memory.cur_capacity} rows x {v.memory.ring.rec_ld * 4} B")
    sched = Schedule(actor, v, p, env, cfg, device, critic, policy, mode=mode)
    devices = (device, v.device, p.device)

    def barrier():
        if pg is not None:
            torch.distributed.barrier(group=pg)

    def timed_block():
        """EXACTLY --steps steps between barrier + device synchronisation on both sides; max over ranks."""
        sched.align()   # (a block starts on a slice boundary: a slice's learner steps are issued at its first step)
        barrier()
        sync_all(*devices)
        t0 = time.perf_counter()
        for i in range(args.steps):
            sched.step(args.steps - i)
        sync_all(*devices)
        barrier()
        dt = time.perf_counter() - t0
        if pg is not None:
            tt = torch.tensor([dt], device=device if args.backend == "nccl" else "cpu", dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX, group=pg)
            dt = float(tt.item())
        return dt

    if args.burn_in_ms > 0:
        t_burn = time.perf_counter()
        while (time.perf_counter() - t_burn) * 1e3 < args.burn_in_ms:
            gemm_section_ms(v, iters=8)   # scratch workspaces only: parameters, Adam state and rings untouched
        note(f"burn-in {args.burn_in_ms:.0f} ms done")
    for i in range(args.warmup):
        sched.step(args.warmup - i)
    sync_all(*devices)
    note("warm-up done")
    dt = timed_block()
    note(f"timed {args.steps} steps in {dt:.3f} s")
    blocks = [dt] + [timed_block() for _ in range(max(args.repeat, 1) - 1)]

    hidden = [int(x) for x in args.hidden.split(",")]
    O, A = env.obs_dim, env.act_dim
    out_c = 51 if args.distl else 1
    f_v, f_p = flops_per_step(O, A, hidden, out_c, args.batch)
    # weak: every rank steps its own batch-B problem -> B-sized gradient steps summed over ranks; strong: one step of the
    # job's batch (B = N x B/N) per schedule step
    per_step = 1 if strong else world
    value = per_step * args.steps / dt
    rates = sorted(per_step * args.steps / t for t in blocks)
    n_gpus = 1 if args.share_gpu else args.gpus
    if split:
        par = "split2 (simulator + rollout policy on GPU 0, V- and P-learner on GPU 1, copy-stream hand-offs over xGMI)"
    elif world > 1:
        par = f"dp{world}"
    else:
        par = "single"
    backend = torch.distributed.get_backend(pg) if pg is not None else None
    unit = {"schedule": "V-learner grad-steps/s", "v_only": "V-learner grad-steps/s", "p_only": "P-learner grad-steps/s"}[mode]
    line = {
        "metric": "learner grad-steps/sec + env-steps/sec, 4096 envs batch 8192 (value = V-learner grad-steps/s at the design schedule "
                  "1 env-iteration : 4 P-steps : 8 V-steps -- critic_sample_ratio 8, critic_actor_ratio 2; weak scaling: batch-sized "
                  "steps summed over ranks, strong scaling: steps of the job's batch)",
        "value": value, "unit": unit, "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"PQL {args.gpus}xMI355X"
                               f"{' per rank (data parallel, ' + ('RCCL' if backend == 'nccl' else str(backend)) + ' grad all-reduce)' if world > 1 else ''}"
                               f"{' functional split' if split else ''}: "
                               f"{args.num_envs} synthetic envs ({args.task}-shape obs={O} act={A}), replay "
                               f"{args.replay} rows resident in HBM, batch {args.batch}, n-step {args.nstep}, "
                               f"{'DistributionalDoubleQ(51)' if args.distl else 'DoubleQ'} MLP {hidden}",
                   "schedule": {"schedule": "1 env-iteration : 4 P-steps : 8 V-steps", "v_only": "v_only", "p_only": "p_only"}[mode],
                   "overrides": list(args.override), "graph": not args.no_graph, "rng": v.rng, "burn_in_ms": args.burn_in_ms, "streams": not args.no_streams, "fused_forward": not args.no_fused,
                   "parallelism": par, "layout": args.layout if args.gpus > 1 else "single", "ranks": world,
                   "per_rank": {"num_envs": args.num_envs, "replay_rows": args.replay, "batch": args.batch},
                   "job": {"num_envs": args.shard.total_envs, "replay_rows": args.replay * world, "batch": args.batch * world},
                   "backend": backend, "share_gpu": bool(args.share_gpu),
                   "grad_buckets": (len(v._buckets) if getattr(v, "_buckets", None) else 1) if getattr(v, "dp", False) else None},
        "repeats": {"blocks": len(blocks), "steps_per_block": args.steps, "median": rates[len(rates) // 2], "min": rates[0],
                    "max": rates[-1], "note": "`value` is the first block; same unit"},
        "p_grad_steps_per_s": value / int(cfg.algo.critic_actor_ratio) if mode == "schedule" else 0.0,
        "env_steps_per_s": (args.steps / dt) / int(cfg.algo.critic_sample_ratio) * args.shard.total_envs if mode == "schedule" else 0.0,
        "gflop_per_v_step": f_v / 1e9, "gflop_per_p_step": f_p / 1e9,
    }
    if rank == 0 and args.no_roofline:
        if args.gather_event_probe:
            K = int(v._workspace(args.batch)["K"])
            t_l, t_n = gather_isolated_ms(v, K, in_step="events")
            print(f"[gather-event-probe] events around one eager launch inside a V step: {t_l:.2f} us; around a one-block no-op: {t_n:.2f} us", file=sys.stderr)
        emit(line)
    elif rank == 0:
        traffic, traffic_src = {}, None
        import glob
        tpaths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))   # committed rocprofv3 --pmc passes; newest tag
        if tpaths and args.hidden == "512,512,256" and args.task == "AllegroHand" and args.batch == 8192:
            traffic = json.load(open(tpaths[-1]))
            traffic_src = os.path.relpath(tpaths[-1], ROOT)
        mfma_util, mfma_src = None, None
        mpaths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_mfma.json")))   # committed matrix-pipe counter pass; newest tag
        if mpaths and traffic_src is not None:
            mfma_util = {k: round(v["mfma_util"], 3) for k, v in json.load(open(mpaths[-1]))["kernels"].items()}
            mfma_src = os.path.relpath(mpaths[-1], ROOT)
        # roofline of the dominant kernel family: the fp32-MFMA GEMMs of one V step (k_gemm<...>)
        ms = gemm_section_ms(v)
        achieved = f_v / (ms * 1e-3) / 1e12
        line["roofline"] = {"bound": "mfma", "kernel": "k_gemm + k_mlp_fwd_fused (all fp32 v_mfma_f32_32x32x2 launches of one V-learner step, and the slab sum that ends the backward call)",
                            "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                            "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic.get("mfma_family_per_v_step_bytes"),
                            "traffic_source": (f"committed profile {traffic_src}, NOT measured in this run: rocprofv3 --pmc FETCH_SIZE / "
                                               "WRITE_SIZE passes of the same command on an earlier box (FETCH_SIZE x2 + WRITE_SIZE at the "
                                               "L2<->fabric boundary, per V step)") if traffic_src else None,
                            "ms_per_launch_group": ms,
                            "mfma_util": mfma_util,
                            "mfma_util_source": (f"committed profile {mfma_src}, NOT measured in this run: SQ_VALU_MFMA_BUSY_CYCLES / "
                                                 "(GRBM_GUI_ACTIVE x 1024 SIMDs) per kernel") if mfma_src else None}
        # (schedule mode only: the --v-only / --p-only runs are the profiler's per-launch-group passes, tools/pmc_traffic.py)
        if mode == "schedule" and v._fused and v.pk_target is not None and v.pk_target.tensor is not None:
            dms = dominant_kernel_ms(v)
            f_dom = 2.0 * args.batch * 2 * mlp_macs([O + A] + hidden + [out_c])   # twin critic, forward only
            line["roofline"]["dominant_kernel"] = {
                "kernel": "k_mlp_fwd_fused (twin-critic forward incl. the Q head, no stash): one launch",
                "us_per_launch": dms * 1e3, "gflop_per_launch": f_dom / 1e9, "achieved": f_dom / (dms * 1e-3) / 1e12,
                "frac": f_dom / (dms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS}
        # the replay gather as the schedule launches it: one launch for the next K V-steps (K = 1 in the per-step torch-RNG mode)
        K = int(v._workspace(args.batch)["K"])
        gms_b2b = gather_ms(v, K)
        alone = getattr(v, "dp", False) or v._ahead is None
        if alone:
            # data parallel: a learner step is collective (the other ranks wait at the barrier below), so rank 0 cannot run its learner
            # alone; algo.rng=torch: the per-step gather sits inside the step's hipGraph, where no event can be recorded around it --
            # the launch is timed one at a time behind a clean read sweep instead (graph differencing)
            g_us, g_rows = gather_isolated_ms(v, K, in_step=False) * 1e3, K * args.batch
        else:
            g_us, g_rows = learner_gather_us(v)
        gms = g_us * 1e-3
        rec_ld = v.memory.ring.rec_ld
        per_row = (2 * O + A) * 4 + 4 + 1 + 8 + (2 * O + A) * 4 + 4 + 4   # 1557 B/sample @cfg2
        per_batch = args.batch * per_row
        alg_bytes = g_rows * per_row
        line["roofline_gather"] = {"bound": "hbm", "kernel": "k_replay_gather_fast (the launch PQLVLearner._prefetch issues: the next K steps' rows)",
                                   "achieved": alg_bytes / (gms * 1e-3) / 1e9,
                                   "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": alg_bytes / (gms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                   "method": ("one launch at a time behind a clean 384-MB read sweep, graph differencing (data parallel or per-step draws: "
                                              "the learner's own launch cannot be bracketed)" if alone else
                                              "kernel duration of the learner's OWN gather launch while it runs its steps as the schedule issues them "
                                              "(learn_many): HIP events right around the launch, minus the same interval around a one-block no-op "
                                              "launch, plus that no-op's 3.4 us (bench.learner_gather_us); = what rocprofv3 --kernel-trace reports"),
                                   "traffic": traffic.get("gather_per_launch_bytes") if traffic.get("gather_batches_per_launch", 1) == K else None,
                                   "traffic_source": (f"committed profile {traffic_src}, NOT measured in this run") if traffic_src else None,
                                   "algorithmic_bytes": alg_bytes,
                                   "us_per_launch": gms * 1e3, "batches_per_launch": g_rows // args.batch, "rows_per_launch": g_rows,
                                   "us_per_batch": gms * 1e3 * args.batch / g_rows, "record_bytes": rec_ld * 4,
                                   "frac_back_to_back": K * per_batch / (gms_b2b * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                   "us_per_launch_back_to_back": gms_b2b * 1e3,
                                   "us_added_to_the_v_queue": gather_isolated_ms(v, K) * 1e3,
                                   "us_added_note": "(graph of 16 x [V step, gather]) - (graph of 16 x [V step]): the launch, its boundary and what "
                                                    "its 110 MB of traffic cost the step's next kernels in evicted weights and activations"}
        if K > 1:   # the per-step launch (algo.rng=torch, injected draws) for comparison
            g1 = gather_isolated_ms(v, 1, in_step=False)
            line["roofline_gather"]["single_batch_launch"] = {"us_per_launch": g1 * 1e3, "frac": per_batch / (g1 * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                                              "method": "one launch at a time behind a clean read sweep, graph differencing"}
        if mode == "schedule" and p.ready_to_learn() and not getattr(p, "dp", False) and p._ahead is not None:   # the P-learner's obs gather: read O floats + 8 B, write O floats per sample
            p_us, prows = learner_gather_us(p)
            pms_b2b, _ = gather_p_ms(p, False)
            p_alg = prows * (2 * O * 4 + 8)
            line["roofline_gather_p"] = {"bound": "hbm", "kernel": "k_replay_gather_obs (the launch PQLPLearner._prefetch issues)",
                                         "achieved": p_alg / (p_us * 1e-6) / 1e9, "peak": PEAK_HBM_GBS,
                                         "unit": "GB/s", "frac": p_alg / (p_us * 1e-6) / 1e9 / PEAK_HBM_GBS, "algorithmic_bytes": p_alg,
                                         "us_per_launch": p_us, "rows_per_launch": prows, "method": "as roofline_gather, on the P-learner's queue",
                                         "frac_back_to_back": p_alg / (pms_b2b * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                         "us_per_launch_back_to_back": pms_b2b * 1e3}
        note("roofline sections measured")
        if world == 1 and mode == "schedule":
            line["free_running"] = free_running(actor, v, p, env, cfg, device)
            line["free_running_concurrent"] = free_running_concurrent(actor, v, p, env, cfg, device, sched.critic, sched.policy)
            note("free-running rates measured")
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args, O, A, hidden)
            note("cpu baseline done")
            line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        emit(line)
    if pg is not None:
        barrier()   # the other ranks wait here while rank 0 measures its roofline sections
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
