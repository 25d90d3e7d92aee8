"""V-learner: the critic side of Parallel Q-Learning on one MI355X.

Drop-in for `pql/algo/pql_v_learner.py`: `PQLVLearner(obs_dim, action_dim, cfg)` with `start()`,
`learn()`, `update(actor, trajectory, normalize_tuple, sleep_time)` and the module-level pump
`asyn_v_learner(learner, cfg)`.  The reference wraps the class in a Ray actor (:21) and ships whole
nn.Modules through the object store; here it is a plain object that owns a HIP stream's worth of work:

learn()  (reference :73-115, ~150 ATen launches + 1 host sync)  ->  one fixed launch sequence
    randint -> fused gather+normalise+cat -> actor fwd (+target-policy noise, written straight into the
    target critic's input) -> target twin-critic fwd -> twin-critic fwd -> TD/MSE or C51/BCE loss + dL/dQ
    -> critic bwd (split-batch dW) -> clip + AdamW + Polyak over flat arenas.
No `.item()`: losses land in a device ring read lazily by `update()`.  With `cfg.algo.graph` the sequence
is captured once into a hipGraph (through torch.cuda.CUDAGraph) and replayed.
"""
from __future__ import annotations

import contextlib
import os
import ctypes as C
import threading
import time
from collections import deque
from copy import deepcopy

import torch

from pql_amd import _lib as L
from pql_amd.models import model_name_to_path
from pql_amd.models.mlp import PackedWeights, default_splits, mlp_forward_raw, output_view
from pql_amd.replay.simple_replay import ReplayBuffer
from pql_amd.utils import dp as DP
from pql_amd.utils import handoff as H
from pql_amd.utils import rng as R
from pql_amd.utils.common import Tracker, load_class_from_path

LOSS_RING = 5  # Tracker(5) of the reference (:54)
# gather: +-5 clamp (bit 0); the learners' input tiles are allocated zeroed and nothing else writes their pad columns (bit 1)
# (PQL_GATHER_FLAGS: A/B switch for the launch-shape / cache-policy bits of include/pqlk.h, tools/ab_bench.sh)
GATHER_FLAGS = int(os.environ.get("PQL_GATHER_FLAGS", 1 | 2))


def _cfg_get(node, name, default=None):
    try:
        v = getattr(node, name)
    except (AttributeError, KeyError):
        return default
    return default if v is None else v


class _AdamState:
    """m, v, step counter and scratch for one parameter arena."""

    def __init__(self, arena: torch.Tensor):
        self.m = torch.zeros_like(arena)
        self.v = torch.zeros_like(arena)
        self.step = torch.zeros(1, dtype=torch.int32, device=arena.device)
        self.gnorm = torch.zeros(1, dtype=torch.float32, device=arena.device)
        self.scratch = torch.zeros(2048, dtype=torch.float32, device=arena.device)


def apply_optimizer(arena, grads, st: _AdamState, target, lr, max_grad_norm, tau, grad_scale=1.0, device=None, layout=None,
                    packed=None, packed_target=None):
    """clip_grad_norm_ + AdamW(torch defaults: betas .9/.999, eps 1e-8, wd 1e-2) + optional Polyak.
    With `layout` + `packed` (PackedWeights with a tensor) the same launch also refreshes the fragment-ordered weight
    copies of the fused forward path."""
    mn = float(max_grad_norm) if max_grad_norm is not None else 0.0
    if layout is not None and packed is not None and packed.tensor is not None:
        pt = packed_target.tensor if packed_target is not None else None
        L.check(L.lib.pqlk_clip_adamw_polyak_pack(C.byref(layout.desc), L.ptr(arena), L.ptr(grads), L.ptr(st.m), L.ptr(st.v),
                                                  L.ptr(target), L.ptr(packed.tensor), L.ptr(pt), float(grad_scale), mn, float(lr),
                                                  0.9, 0.999, 1e-8, 1e-2, float(tau), L.ptr(st.step), L.ptr(st.gnorm),
                                                  L.ptr(st.scratch), L.stream(device)))
        return
    L.check(L.lib.pqlk_clip_adamw_polyak(L.ptr(arena), L.ptr(grads), L.ptr(st.m), L.ptr(st.v), L.ptr(target),
                                         arena.numel(), float(grad_scale),
                                         float(max_grad_norm) if max_grad_norm is not None else 0.0,
                                         float(lr), 0.9, 0.999, 1e-8, 1e-2, float(tau), L.ptr(st.step), L.ptr(st.gnorm),
                                         L.ptr(st.scratch), L.stream(device)))


def apply_optimizer_fused(layout, arena, grads, st: _AdamState, target, lr, max_grad_norm, tau, packed, packed_target,
                          loss_part, loss_parts, loss_scale, loss_ring, device, norm_in_backward=True, grad_scale=1.0):
    """Tail of a fused learner step: AdamW (+ Polyak + re-pack) whose launch also folds the loss partials into the loss ring.
    norm_in_backward (single GPU): the squared-norm partials and the step increment were left in `st.scratch` / `st.step` by
    `pqlk_mlp_backward_norm` -- same bits as apply_optimizer + the stand-alone folds, two launches fewer.  Data parallel
    (norm_in_backward=False, grad_scale = 1 / world): the norm pass runs here, on the all-reduced gradient; one launch fewer."""
    mn = float(max_grad_norm) if max_grad_norm is not None else 0.0
    pk = packed.tensor if packed is not None else None
    pt = packed_target.tensor if packed_target is not None else None
    L.check(L.lib.pqlk_adamw_polyak_fused(C.byref(layout.desc), L.ptr(arena), L.ptr(grads), L.ptr(st.m), L.ptr(st.v), L.ptr(target),
                                          L.ptr(pk), L.ptr(pt), float(grad_scale), mn, float(lr), 0.9, 0.999, 1e-8, 1e-2, float(tau),
                                          L.ptr(st.step), L.ptr(st.gnorm), L.ptr(st.scratch),
                                          int(L.lib.pqlk_mlp_norm_parts(C.byref(layout.desc))) if norm_in_backward else 0,
                                          L.ptr(loss_part), int(loss_parts),
                                          float(loss_scale), L.ptr(loss_ring), LOSS_RING, L.stream(device)))


def f32_recip(*factors, sign=1.0):
    """sign / (f0 * f1 ...) evaluated in fp32, like the kernels' `1.0f / ((float)b * (float)k)`."""
    import numpy as np
    d = np.float32(1.0)
    for f in factors:
        d = np.float32(d * np.float32(f))
    return float(np.float32(sign) / d)


def allreduce_sum(t, pg):
    """Sum-all-reduce of the flat gradient arena.  RCCL ("nccl") reduces in place on the device over xGMI; the
    gloo rehearsal path (CPU tests / one-GPU dry runs) stages through host memory."""
    if torch.distributed.get_backend(pg) == "gloo" and t.is_cuda:
        h = t.cpu()
        torch.distributed.all_reduce(h, group=pg)
        t.copy_(h)
    else:
        torch.distributed.all_reduce(t, group=pg)


class LaggedLoss:
    """Mean of the last LOSS_RING losses without stalling the stream: each call enqueues an async copy of the
    device ring to pinned host memory and returns the value of the last copy that has completed (one hand-off
    behind).  Replaces the reference's per-step `loss.item()` + Tracker(5) (pql_v_learner.py:111)."""

    def __init__(self, ring: torch.Tensor):
        self.ring = ring
        self.host = torch.zeros(ring.numel(), dtype=torch.float32).pin_memory()
        self.event = None
        self.count_at_copy = 0
        self.value = 0.0

    @staticmethod
    def mean_of(vals, count):
        n = min(count, LOSS_RING)
        window = [vals[t % LOSS_RING] for t in range(count - n, count)]
        return float(sum(window) / LOSS_RING)   # Tracker(5) is zero-filled: always divides by its length

    def poll(self, count):
        if self.event is not None and self.event.query():
            self.value = self.mean_of(self.host.tolist(), self.count_at_copy)
            self.event = None
        if self.event is None:
            self.host.copy_(self.ring, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()
            self.count_at_copy = count
        return self.value


def graph_collective_enabled(pg):
    """PQL_DP_GRAPH_COLLECTIVE=1 captures the gradient all-reduce inside the learner's hipGraph.  Only RCCL can be
    captured (the gloo rehearsal path stages through the host); rehearsed with a 1-rank group only -- unverified for
    world > 1, scaling was not measurable on this pool."""
    if os.environ.get("PQL_DP_GRAPH_COLLECTIVE", "0") != "1":
        return False
    if torch.distributed.get_backend(pg) != "nccl":
        raise L.PqlkError("PQL_DP_GRAPH_COLLECTIVE=1 needs the RCCL ('nccl') backend: a gloo all-reduce cannot be graph-captured")
    return True


def resident_norm(owner, normalize_tuple, home=None):
    """Copy (mean, var, eps) into buffers that live as long as the learner, so kernels (and captured
    graphs) always read the same addresses; the producer may hand over fresh tensors every iteration.
    Runs on the learner's stream (current), fenced against the stream the tensors were produced on."""
    if normalize_tuple is None:
        return None
    mean, var, eps = normalize_tuple
    cur = getattr(owner, "_norm_buf", None)
    if cur is None or cur[0].shape != mean.reshape(-1).shape:
        cur = (torch.empty(mean.numel(), dtype=torch.float32, device=owner.device),
               torch.empty(var.numel(), dtype=torch.float32, device=owner.device))
        owner._norm_buf = cur
    st = torch.cuda.current_stream(owner.device)
    for dst, src in zip(cur, (mean, var)):
        with H.LOCK:
            lease = H.acquire(src, st, home)
            dst.copy_(src.reshape(-1), non_blocking=True)
            H.release(lease, st)
    return cur[0], cur[1], float(eps)


def adopt_arena(dst_module, src_module, device, home=None, pipe="params"):
    """Fenced copy of `src_module`'s flat arena into `dst_module`'s on the CURRENT stream of `device`: waits for the
    producer (a published snapshot's event, or the caller's stream for a plain module) and releases the source
    afterwards.  From another GPU the bytes first land in a double-buffered block through the copy streams (peer copy
    over xGMI), so the learner's compute stream only ever does the local arena copy."""
    st = torch.cuda.current_stream(device)
    with H.LOCK:
        if H.crosses(src_module.arena.device, device):
            blk = H.shipper(src_module.arena.device, device, pipe).ship((src_module.arena.data,), H.lease_of(src_module))
            lease = H.acquire(blk, st)
            dst_module.arena.data.copy_(blk[0], non_blocking=True)
        else:
            lease = H.acquire(src_module, st, home)
            dst_module.arena.data.copy_(src_module.arena.data, non_blocking=True)
        H.release(lease, st)


def pump(learner, stop_event=None, max_in_flight=2):
    """Free-running learner loop shared by asyn_v_learner / asyn_p_learner.  The host enqueues a step in ~15 us and the
    GPU takes ~0.7 ms to run it, so without back-pressure the queue would run thousands of steps ahead of the device and
    every `update()` would land behind them: at most `max_in_flight` steps are kept enqueued (event wait, GIL released)."""
    pending = deque()
    while stop_event is None or not stop_event.is_set():
        if not learner.ready_to_learn():
            time.sleep(0.0005)
            continue
        sleep_time = learner.learn()
        pending.append(learner.fence())
        while len(pending) > max_in_flight:
            pending.popleft().synchronize()
        if sleep_time:
            time.sleep(sleep_time)
    while pending:
        pending.popleft().synchronize()


class PQLVLearner:
    def __init__(self, obs_dim, action_dim, cfg, process_group=None):
        self.cfg = cfg
        self.obs_dim = obs_dim
        self.action_dim = int(action_dim)
        if not torch.cuda.is_available():
            raise L.PqlkError("PQLVLearner needs an MI355X (no CPU path)")
        self.device = torch.device(f"cuda:{int(cfg.algo.v_learner_gpu)}")
        self.pg = process_group  # data-parallel group (RCCL); None = single GPU
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        # dp: the collective is issued even for a 1-rank group, so the RCCL path can be rehearsed on one GPU
        self.dp = process_group is not None

        algo = cfg.algo
        if algo.distl and "Distributional" not in algo.cri_class:
            algo.cri_class = "Distributional" + algo.cri_class  # same rewrite as the reference (:30-31)
        cri_class = load_class_from_path(algo.cri_class, model_name_to_path[algo.cri_class])
        hidden = _cfg_get(algo, "hidden_layers")
        hidden = list(hidden) if hidden is not None else None
        with torch.cuda.device(self.device):
            if algo.distl:
                self.critic = cri_class(self.obs_dim, self.action_dim, v_min=algo.v_min, v_max=algo.v_max,
                                        num_atoms=algo.num_atoms, device=self.device, hidden_layers=hidden).to(self.device)
            else:
                self.critic = cri_class(self.obs_dim, self.action_dim, hidden_layers=hidden).to(self.device)
        if cfg.artifact is not None:
            raise NotImplementedError("W&B artifact download is out of scope (no network); load a local state_dict instead")
        self.critic_target = deepcopy(self.critic)
        self.opt = _AdamState(self.critic.arena.data)
        fused = bool(_cfg_get(algo, "fused", True))
        self.pk_critic = PackedWeights(self.critic.layout, self.device) if fused else None
        self.pk_target = PackedWeights(self.critic.layout, self.device) if fused else None
        self.pk_actor = None
        self._fused = fused
        self._fold_loss = bool(_cfg_get(algo, "fused_tail", True))   # loss partials folded by the optimiser launch
        self._fused_tail = not self.dp and self._fold_loss            # ... and the gradient norm's partials by backward's reduction
        self._td_in_head = bool(_cfg_get(algo, "td_in_head", True))   # TD target + MSE inside the head's backward launch
        self.actor = None
        self.memory = ReplayBuffer(capacity=int(algo.memory_size), obs_dim=self.obs_dim, action_dim=self.action_dim,
                                   device=self.device)
        self.loss_tracker = Tracker(LOSS_RING)
        self.loss_ring = torch.zeros(LOSS_RING, dtype=torch.float32, device=self.device)
        self._lagged = LaggedLoss(self.loss_ring)
        self.update_count = 0
        self.normalize_tuple = None
        self.sleep_time = 0
        self.use_graph = bool(_cfg_get(algo, "graph", False))
        # data parallel, algo.dp_buckets: "layer" = the gradient travels in per-layer buckets, each all-reduced as soon as its dW
        # slabs are summed, under the MFMA launches of the layers below (pql_amd/utils/dp.py); "one" = a single collective after
        # the whole backward; "auto" (default) = layer when the collectives are captured inside the step's hipGraph
        # (PQL_DP_GRAPH_COLLECTIVE=1), else one: with eager collectives every bucket ends a graph, and on the one rank this pool
        # can run those extra graph boundaries cost more (1114 vs 1164 steps/s) than a 1-rank collective can give back
        self._buckets = None
        if self.dp and self.critic.layout.n_layers >= 3:
            mode = str(_cfg_get(algo, "dp_buckets", "auto"))
            if mode not in ("auto", "layer", "one"):
                raise ValueError(f"algo.dp_buckets must be auto, layer or one, got {mode!r}")
            if mode == "auto":
                mode = "layer" if self.use_graph and graph_collective_enabled(self.pg) else "one"
            if mode == "layer":
                self._buckets = DP.layer_buckets(self.critic.layout.n_layers)
                self._reducer = DP.BucketAllReduce(self.pg)
        # RNG draws inside the hipGraph or in front of it.  In front (default): torch hands a captured generator its seed and
        # Philox offset through two 1-element fill launches per replay (~9 us of device time per step, more than the draws save
        # by being captured), and the graph no longer bakes in the randint bound, so it is not re-captured while the ring fills.
        self._graph_rng = bool(_cfg_get(algo, "graph_rng", False))
        # own HIP stream: the MI355X form of the reference's separate learner process (Ray actor).  V-learner,
        # P-learner and rollout queues then overlap on the GPU; hand-offs are event-fenced in update().
        self.stream = torch.cuda.Stream(self.device) if bool(_cfg_get(algo, "streams", False)) else None
        # what start()/update() hand out: double-buffered snapshots of the critic (the reference returns a pickled copy
        # through Ray, pql_v_learner.py:59-60,122), so a consumer never reads an arena AdamW is writing and the weights a
        # caller holds are those of the hand-off, not of whenever it gets round to using them
        self._pub = H.ArenaPublisher(self.critic)
        self._lock = threading.RLock()   # learn() / update() are FIFO like calls on a Ray actor
        self._capture_stream = torch.cuda.Stream(self.device)   # torch's default capture stream is shared by every graph
        # Own device generator, like the reference's learner PROCESS has its own default generator.  Not a
        # nicety: every hipGraph that draws from a generator is handed its Philox offset through ONE device word per
        # generator, refreshed on the replaying stream -- two learners replaying graphs on two streams off the shared
        # default generator overwrite each other's offset (measured: different sample indices from run to run).
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(torch.randint(0, 2 ** 62, (1,)).item()))   # derived from the driver's seed (CPU generator)
        # algo.rng: "auto" (default) = this library produces the draws itself, `algo.prefetch_steps` steps ahead in one launch,
        # together with ONE batched replay gather for those steps -- if its numbers are torch's on this device
        # (pql_amd/utils/rng.py), else "torch"; "torch" = one randint + one normal_ ATen launch in front of every step (round 2);
        # "philox" = as auto, but refuse to run when the check fails.
        self._rng_mode = str(_cfg_get(algo, "rng", "auto"))
        if self._rng_mode != "torch":   # the on-device check runs HERE, once, under a lock (not lazily inside the first learn(),
            with torch.cuda.device(self.device):   # which free-running learners reach from two threads at the same time)
                R.verified(self.device)
        self._depth = max(1, int(_cfg_get(algo, "prefetch_steps", _cfg_get(algo, "critic_sample_ratio", 8))))
        self._ahead = None
        self._ws = None
        self._graph = None
        self._graph_post = None
        self._graph_key = None
        self._slot_graphs = {}
        self._run_graph = None   # all K draws-ahead steps of one run in ONE hipGraph (learn_many)
        self._run_graphs = bool(_cfg_get(algo, "run_graph", True))

    # ------------------------------------------------------------------------------------------
    def start(self):
        with self._lock, torch.cuda.device(self.device), self._on_stream():
            return self._published(), self.update_count, self.loss_tracker.mean()

    def _on_stream(self):
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def _published(self):
        """The critic as handed to other components: a snapshot taken on this learner's queue."""
        return self._pub.publish()

    def use_private_rng(self, seed):
        """Re-seed this learner's generator."""
        self.gen.manual_seed(int(seed))
        self._graph = None
        self._drop_ahead()

    def _drop_ahead(self):
        """Forget the draws / gathered tiles prepared for later steps (the ring, its bound, the statistics or the generator
        changed): the next step prepares them again at the generator's current offset."""
        if self._ahead is not None:
            self._ahead.invalidate()

    def _data_stamp(self):
        """What the tiles gathered ahead depend on besides the draws: the ring's contents (insert counter), the randint bound and
        the identity of the normalisation statistics.  `update()` drops the tiles itself; this catches every OTHER way the data can
        change under a learner that has steps prepared -- `memory.add_to_buffer(...)` called directly, `normalize_tuple` or
        `memory.cur_capacity` assigned from outside (tools/gen_golden-style drivers, tests) -- which would otherwise train up to
        K - 1 steps on stale rows, stale statistics or a stale bound without any error."""
        nt = self.normalize_tuple
        # (+ the target policy, whose actions for the prepared steps are computed at prefetch time: object and in-place version)
        pol = None if self.actor is None else (id(self.actor), self.actor.arena.data._version)
        return (self.memory.ring.version, self.memory.cur_capacity, None if nt is None else (id(nt[0]), id(nt[1]), float(nt[2])), pol)

    def _norm_key(self):
        """Part of every graph key: a captured gather has the ADDRESSES of the statistics baked in (update() keeps them stable by
        copying into resident buffers; a tuple assigned from outside brings new ones and must re-capture)."""
        nt = self.normalize_tuple
        return None if nt is None else (nt[0].data_ptr(), nt[1].data_ptr(), float(nt[2]))

    def _check_ahead(self):
        if self._ahead is not None and self._ahead.valid and getattr(self, "_ahead_stamp", None) != self._data_stamp():
            self._drop_ahead()

    @property
    def rng(self):
        """'philox' when the draws come from this library's launch, 'torch' when from ATen's (see __init__)."""
        return "philox" if self._ahead is not None else "torch"

    def ready_to_learn(self):
        return self.actor is not None

    def fence(self):
        """Event behind everything enqueued on this learner's queue so far."""
        ev = torch.cuda.Event()
        ev.record(self.stream if self.stream is not None else torch.cuda.current_stream(self.device))
        return ev

    def synchronize(self):
        self.fence().synchronize()

    def _workspace(self, B):
        if self._ws is not None and self._ws["B"] == B:
            return self._ws
        dev, f = self.device, dict(dtype=torch.float32, device=self.device)
        O, A = self.memory.ring.O, self.action_dim
        cl, al = self.critic.layout, self.actor.layout
        # leading dimension of the critic's input tiles: ld(O + A), and beyond 128 floats a multiple of 128, so that the layer-1 dW
        # product (X = these tiles) can read whole 128-column tiles of them on the LDS-DMA loop even when ld(O + A) is not a multiple of
        # the tile (Humanoid: 129 inputs -> ld 160 -> tiles 256 wide; the extra columns are never written and stay zero)
        ld_sa = L.ld(O + A)
        if ld_sa > 128:
            ld_sa = (ld_sa + 127) // 128 * 128
        ws = dict(B=B, ld_sa=ld_sa, ld_o=L.ld(O))
        # Draws and gathered input tiles of the next K steps (K = 1 without the fused draws): `x_sa` / `xn_sa` / `rew` / `done`
        # are slot 0, the tiles of the per-step path.  cfg #2: K = 8 -> 2 x 33.5 MB of tiles.
        want = self._want_ahead(B)
        K = self._depth if want else 1
        self._ahead = R.DrawAhead(self.gen, dev, B, (B, A), K, R.verified(dev)) if want else None
        self._slot_graphs, self._run_graph = {}, None
        ws["K"] = K
        ws["x_sa_all"] = torch.zeros((K, B, ws["ld_sa"]), **f)
        ws["xn_sa_all"] = torch.zeros((K, B, ws["ld_sa"]), **f)
        ws["rew_all"] = torch.zeros((K, B), **f)
        ws["done_all"] = torch.zeros((K, B), **f)
        ws["slots"] = [dict(x_sa=ws["x_sa_all"][k], xn_sa=ws["xn_sa_all"][k], rew=ws["rew_all"][k], done=ws["done_all"][k])
                       for k in range(K)]
        ws.update(ws["slots"][0])
        ws["xn_obs"] = torch.zeros((B, ws["ld_o"]), **f)
        ws["idx"] = torch.zeros(B, dtype=torch.int64, device=dev)
        ws["draw"] = torch.zeros((B, A), **f)
        ws["acts_a"] = torch.empty(al.acts_floats(B), **f)
        # The target policy does not change between two hand-offs (pql_v_learner.py:117-122 is the only place the reference assigns
        # it), its inputs for the next K steps are the tiles gathered ahead and its noise the draws made ahead: the K steps' target
        # actions come from ONE forward launch over K x B rows at prefetch time (64-row tiles on every CU, one launch's fixed cost
        # instead of K) and the step itself starts at the target critic (algo.actor_ahead).
        ws["actor_ahead"] = bool(want and K > 1 and _cfg_get(self.cfg.algo, "actor_ahead", True))
        if ws["actor_ahead"]:
            ws["a_out_all"] = torch.empty((K * B, L.ld(A)), **f)
        ws["acts_t"] = torch.empty(cl.acts_floats(B), **f)
        ws["acts_c"] = torch.empty(cl.acts_floats(B), **f)
        ws["dy"] = torch.zeros((2, B, cl.ld_out), **f)
        ws["grads"] = torch.zeros(cl.total, **f)
        ws["splits"] = default_splits(B, _cfg_get(self.cfg.algo, "dw_splits", 16))
        ws["bwd"] = torch.empty(cl.bwd_ws_floats(B, ws["splits"]), **f)
        ws["scratch"] = torch.zeros(2048, **f)
        # scalar twin heads: TD target + MSE + dL/dQ are formed inside the head's backward pass (one launch less)
        ws["td_parts"] = int(L.lib.pqlk_td_head_loss_parts(C.byref(cl.desc), B)) if (self._fold_loss and self._td_in_head
                                                                                      and not self.cfg.algo.distl) else 0
        # ... and with the fused forward the head's whole backward runs inside the critic's forward launch, off the activations still
        # in LDS: the head-backward launch and its second read of them disappear (algo.td_in_forward; not with gradient buckets)
        ws["td_fwd"] = 0
        if ws["td_parts"] > 0 and bool(_cfg_get(self.cfg.algo, "td_in_forward", True)) and self._buckets is None \
                and self.pk_critic is not None and self.pk_critic.tensor is not None:
            ws["td_fwd"] = int(L.lib.pqlk_td_forward_loss_parts(C.byref(cl.desc), B))
            if ws["td_fwd"] > 0:
                ws["td_parts"] = ws["td_fwd"]
        if ws["td_parts"] > ws["scratch"].numel():   # (batches past 65 536: one loss partial per row tile and net)
            ws["scratch"] = torch.zeros(ws["td_parts"], **f)
        if self._buckets is not None:
            ws["bucket_views"] = [DP.bucket_views(ws["grads"], cl, hi, lo) for hi, lo in self._buckets]
        self._ws = ws
        self.repack()
        return ws

    def _want_ahead(self, B):
        """Fused draws + batched gather: needs the fused actor forward (it reads norm(next_obs) out of the target critic's input
        tile, so a step's inputs are exactly two tiles), draws outside the graphs, and torch's numbers reproduced on this device."""
        if self._rng_mode == "torch" or self._graph_rng or self.pk_actor is None or self.pk_actor.tensor is None:
            return False
        ok = R.verified(self.device) is not None
        if not ok and self._rng_mode == "philox":
            raise L.PqlkError("algo.rng=philox: pqlk_philox_draws does not reproduce torch.randint / normal_ on this device "
                              "(another torch / rocRAND build?); use algo.rng=auto or torch")
        return ok

    def repack(self):
        """Re-derive the fragment-ordered weight copies from the arenas (after loading a state_dict etc.)."""
        if self._fused:
            self.pk_critic.refresh(self.critic.arena.data)
            self.pk_target.refresh(self.critic_target.arena.data)
            if self.pk_actor is not None:
                self.pk_actor.refresh(self.actor.arena.data)

    def _norm_ptrs(self):
        if not self.cfg.algo.obs_norm or self.normalize_tuple is None:
            return None, None, 0.0
        mean, var, eps = self.normalize_tuple
        return mean, var, float(eps)

    def _gather(self, ws, idx, rows, x_sa, xn_sa, rew, done):
        """Fused replay gather (+ normalise + concat) of `rows` samples into the given tiles."""
        mean, var, eps = self._norm_ptrs()
        # the fused actor forward masks everything past column O while staging its tile, so it can read norm(next_obs)
        # straight out of the target critic's input tile: one gather output (B x ld(O) floats) less to write
        actor_in_sa = self.pk_actor is not None and self.pk_actor.tensor is not None
        L.check(L.lib.pqlk_replay_gather_fused(C.byref(self.memory.ring.desc), L.ptr(idx), rows, L.ptr(mean), L.ptr(var), eps, GATHER_FLAGS,
                                               L.ptr(x_sa), ws["ld_sa"], L.ptr(xn_sa), None if actor_in_sa else L.ptr(ws["xn_obs"]),
                                               ws["ld_o"], L.ptr(rew), L.ptr(done), L.stream(self.device)))

    def _prefetch(self, ws, steps=None):
        """Draws of the next K steps in one launch (torch's own numbers, pql_amd/utils/rng.py) and ONE gather of their K x B
        rows: the ring does not change between two `update()` calls, so what the reference samples at the start of each of
        those steps (simple_replay.py:85-104) can be fetched together -- 102 MB per launch at cfg #2 instead of eight
        latency-bound 12.75-MB launches.  `steps` < K (learn_many of a partial run): only that many steps' draws, rows and
        target actions -- the rest would be dropped unused at the next `update()`."""
        K, B = ws["K"], ws["B"]
        Kp = K if steps is None else max(1, min(K, int(steps)))
        self._ahead.refill(self.memory.cur_capacity, Kp)
        self._gather(ws, self._ahead.idx, Kp * B, ws["x_sa_all"], ws["xn_sa_all"], ws["rew_all"], ws["done_all"])
        if ws["actor_ahead"]:   # a' = clamp(tanh(actor(s')) + clamp(0.8 N(0,1), +-0.2), +-1) of all K steps -> action columns of their tiles
            algo, O = self.cfg.algo, self.memory.ring.O
            xn = ws["xn_sa_all"].view(K * B, ws["ld_sa"])[: Kp * B]
            mlp_forward_raw(self.actor.layout, self.actor.arena.data, xn, L.ACT_TANH_NOISE, self._ahead.normal.view(K * B, -1)[: Kp * B],
                            algo.noise.tgt_pol_std, algo.noise.tgt_pol_noise_bound, ws["a_out_all"], xn[:, O:], packed=self.pk_actor, stash_all=2)
        self._ahead_stamp = self._data_stamp()

    def _step_kernels(self, ws, idx, draw, upto_backward=False, tiles=None, part=None):
        """The launch sequence of one critic gradient step; everything asynchronous on the current stream.
        upto_backward=True stops after the gradient is formed (graph capture around the DP all-reduce).
        tiles: input tiles already gathered by `_prefetch` (a slot of ws["slots"]); None = gather `idx` into slot 0 here.
        part (data-parallel buckets, graph capture): only the forward passes + bucket 0 (part = 0) or bucket `part` of the
        backward, no collective."""
        algo, dev, B = self.cfg.algo, self.device, ws["B"]
        O = self.memory.ring.O
        st = L.stream(dev)
        actor_in_sa = self.pk_actor is not None and self.pk_actor.tensor is not None
        tiles_ahead = tiles is not None
        if tiles is None:
            tiles = ws["slots"][0]
            if part in (None, 0):
                self._gather(ws, idx, B, tiles["x_sa"], tiles["xn_sa"], tiles["rew"], tiles["done"])
        ws = dict(ws, **tiles)   # the step below reads its inputs from `tiles`
        al, cl = self.actor.layout, self.critic.layout
        gamma_n = float(algo.gamma) ** int(algo.nstep)
        # single GPU: the loss fold and the gradient-norm pass ride in launches that exist anyway (backward's slab
        # reduction, the optimiser); data parallel keeps them apart because the all-reduce sits in between
        tail = self._fused_tail
        if part in (None, 0):
            # target policy smoothing (:63-71): a' written into the action columns of the target critic's input.
            # The two no-grad chains (actor, target critic) skip the activation stash; the critic keeps it for backward.
            xn_act = ws["xn_sa"][:, O:]
            if not (tiles_ahead and ws["actor_ahead"]):   # (tiles gathered ahead already hold a': _prefetch)
                mlp_forward_raw(al, self.actor.arena.data, ws["xn_sa"] if actor_in_sa else ws["xn_obs"], L.ACT_TANH_NOISE, draw, algo.noise.tgt_pol_std,
                                algo.noise.tgt_pol_noise_bound, ws["acts_a"], xn_act, packed=self.pk_actor, stash_all=False)
            mlp_forward_raw(cl, self.critic_target.arena.data, ws["xn_sa"], L.ACT_NONE, acts=ws["acts_t"], packed=self.pk_target,
                            stash_all=False)
            if ws["td_fwd"] > 0:
                L.check(L.lib.pqlk_mlp_forward_td(C.byref(cl.desc), L.ptr(self.critic.arena.data), L.ptr(self.pk_critic.tensor), L.ptr(ws["x_sa"]),
                                                  ws["ld_sa"], B, L.ptr(ws["acts_c"]), L.ptr(ws["acts_t"]), L.ptr(ws["rew"]), L.ptr(ws["done"]),
                                                  gamma_n, L.ptr(ws["scratch"]), L.ptr(ws["bwd"]), ws["bwd"].numel(), ws["splits"], st))
            else:
                mlp_forward_raw(cl, self.critic.arena.data, ws["x_sa"], L.ACT_NONE, acts=ws["acts_c"], packed=self.pk_critic,
                                stash_all=True)
            if ws["td_parts"] == 0:
                q = output_view(cl, ws["acts_c"], B)
                qt = output_view(cl, ws["acts_t"], B)
                loss_out = None if self._fold_loss else L.ptr(self.loss_ring)
                if algo.distl:
                    L.check(L.lib.pqlk_c51_bce_loss(L.ptr(q), L.ptr(qt), cl.ld_out, int(algo.num_atoms), L.ptr(ws["rew"]),
                                                    L.ptr(ws["done"]), L.ptr(self.critic.z_atoms), gamma_n, float(algo.v_min),
                                                    float(algo.v_max), B, L.ptr(ws["dy"]), loss_out, L.ptr(self.opt.step),
                                                    LOSS_RING, None, L.ptr(ws["scratch"]), st))
                else:
                    L.check(L.lib.pqlk_td_mse_loss(L.ptr(q), L.ptr(qt), cl.ld_out, L.ptr(ws["rew"]), L.ptr(ws["done"]), gamma_n, B,
                                                   L.ptr(ws["dy"]), loss_out, L.ptr(self.opt.step), LOSS_RING,
                                                   L.ptr(ws["scratch"]), st))
        if self._buckets is not None:
            # data parallel: the chain in pieces, each ending with the sum of its layers' dW slabs; that piece's collective goes
            # out right behind it and runs under the launches of the layers below
            td = ws["td_parts"] > 0
            for k, (hi, lo) in enumerate(self._buckets):
                if part is not None and part != k:
                    continue
                L.check(L.lib.pqlk_mlp_backward_layers(C.byref(cl.desc), L.ptr(self.critic.arena.data), L.ptr(ws["x_sa"]), ws["ld_sa"], B,
                                                       L.ptr(ws["acts_c"]), None if td else L.ptr(ws["dy"]),
                                                       L.ptr(ws["acts_t"]) if td else None, L.ptr(ws["rew"]) if td else None,
                                                       L.ptr(ws["done"]) if td else None, gamma_n, L.ptr(ws["scratch"]) if td else None,
                                                       L.ptr(ws["grads"]), ws["splits"], L.ptr(ws["bwd"]), ws["bwd"].numel(), hi, lo, st))
                if part is None:
                    self._reducer.issue(ws["bucket_views"][k])
            if part is not None:
                return
            self._reducer.wait()
            self._step_post(ws)
            return
        if ws["td_fwd"] > 0:
            L.check(L.lib.pqlk_mlp_backward_td_tail(C.byref(cl.desc), L.ptr(self.critic.arena.data), L.ptr(ws["x_sa"]), ws["ld_sa"], B,
                                                    L.ptr(ws["acts_c"]), L.ptr(ws["grads"]), ws["splits"], L.ptr(ws["bwd"]), ws["bwd"].numel(),
                                                    L.ptr(self.opt.scratch) if tail else None, L.ptr(self.opt.step) if tail else None, st))
        elif ws["td_parts"] > 0:
            L.check(L.lib.pqlk_mlp_backward_td(C.byref(cl.desc), L.ptr(self.critic.arena.data), L.ptr(ws["x_sa"]), ws["ld_sa"], B,
                                               L.ptr(ws["acts_c"]), L.ptr(ws["acts_t"]), L.ptr(ws["rew"]), L.ptr(ws["done"]), gamma_n,
                                               L.ptr(ws["scratch"]), L.ptr(ws["grads"]), ws["splits"], L.ptr(ws["bwd"]),
                                               ws["bwd"].numel(), L.ptr(self.opt.scratch) if tail else None,
                                               L.ptr(self.opt.step) if tail else None, st))
        elif tail:
            L.check(L.lib.pqlk_mlp_backward_norm(C.byref(cl.desc), L.ptr(self.critic.arena.data), L.ptr(ws["x_sa"]), ws["ld_sa"], B,
                                                 L.ptr(ws["acts_c"]), L.ptr(ws["dy"]), L.ptr(ws["grads"]), ws["splits"], None, 0, 0,
                                                 0, None, 0, L.ptr(ws["bwd"]), ws["bwd"].numel(), L.ptr(self.opt.scratch),
                                                 L.ptr(self.opt.step), st))
        else:
            L.check(L.lib.pqlk_mlp_backward(C.byref(cl.desc), L.ptr(self.critic.arena.data), L.ptr(ws["x_sa"]), ws["ld_sa"], B,
                                            L.ptr(ws["acts_c"]), L.ptr(ws["dy"]), L.ptr(ws["grads"]), ws["splits"], None, 0, 0, 0,
                                            None, 0, L.ptr(ws["bwd"]), ws["bwd"].numel(), st))
        if upto_backward:
            return
        self._allreduce_grads(ws)
        self._step_post(ws)

    def _allreduce_grads(self, ws):
        if self.dp:  # data-parallel: ONE collective per step, sum over ranks on RCCL; the mean is folded into
            allreduce_sum(ws["grads"], self.pg)                         # the optimiser's grad_scale

    def _step_post(self, ws):
        algo, dev = self.cfg.algo, self.device
        if self._fold_loss:
            K = int(algo.num_atoms) if algo.distl else 1
            apply_optimizer_fused(self.critic.layout, self.critic.arena.data, ws["grads"], self.opt, self.critic_target.arena.data,
                                  algo.critic_lr, algo.max_grad_norm, algo.tau, self.pk_critic, self.pk_target, ws["scratch"],
                                  ws["td_parts"] or L.lib.pqlk_loss_parts(ws["B"], K), f32_recip(ws["B"], K) if K > 1 else f32_recip(ws["B"]),
                                  self.loss_ring, dev, norm_in_backward=self._fused_tail, grad_scale=1.0 / self.world)
            return
        # optimiser + Polyak + refresh of the fragment-ordered weight copies (critic and target) in one launch pair
        apply_optimizer(self.critic.arena.data, ws["grads"], self.opt, self.critic_target.arena.data, algo.critic_lr,
                        algo.max_grad_norm, algo.tau, 1.0 / self.world, dev, layout=self.critic.layout,
                        packed=self.pk_critic, packed_target=self.pk_target)

    def _draws(self, ws):
        # RNG consumption order of the reference: one randint(cur_capacity,(B,)) then one
        # N(0,1) draw of shape (B, A) on the learner's device generator.
        torch.randint(self.memory.cur_capacity, (ws["B"],), generator=self.gen, out=ws["idx"])   # straight into the workspace: no copy launch
        ws["draw"].normal_(generator=self.gen)

    def _draw_and_step(self, ws, upto_backward=False, draw=True, part=None):
        if draw:
            self._draws(ws)
        self._step_kernels(ws, ws["idx"], ws["draw"], upto_backward, part=part)

    @torch.no_grad()
    def learn(self, indices=None, noise=None):
        """One critic gradient step.  `indices` (B,) int64 and `noise` (B,A) N(0,1) draws may be injected
        for parity tests; otherwise they are drawn exactly like the reference draws them."""
        if self.actor is None:
            return self.sleep_time
        B = int(self.cfg.algo.batch_size)
        home = torch.cuda.current_stream(self.device)
        with self._lock, torch.cuda.device(self.device), self._on_stream():
            ws = self._workspace(B)
            if indices is not None or noise is not None:
                if indices is None or noise is None:
                    self._drop_ahead()   # the generator is about to be used directly: what was drawn ahead is off the stream now
                if indices is not None:   # injected draws arrive on the caller's stream (or from the host)
                    self._inject(ws["idx"], indices, home)
                else:
                    torch.randint(self.memory.cur_capacity, (B,), generator=self.gen, out=ws["idx"])
                if noise is not None:
                    self._inject(ws["draw"], noise, home)
                else:
                    ws["draw"].normal_(generator=self.gen)
                self._step_kernels(ws, ws["idx"], ws["draw"])
            elif self._ahead is not None and self.memory.cur_capacity < (1 << 28):
                # draws + input tiles of the next K steps come from one launch pair (`_prefetch`), the step itself has no RNG
                # and no gather launch left; one hipGraph per slot (the tiles' addresses are baked in)
                self._check_ahead()
                if self._ahead.valid == 0:
                    self._prefetch(ws)
                slot = self._ahead.take()
                if self.use_graph:
                    key = (B, 0, id(self.actor), self._norm_key())
                    if self._graph_key != key:
                        self._slot_graphs, self._run_graph, self._graph, self._graph_post, self._graph_key = {}, None, None, None, key
                    if slot not in self._slot_graphs:
                        with H.CAPTURE_LOCK:
                            self._capture(ws, key, slot)
                    self._replay(ws, self._slot_graphs[slot])
                else:
                    self._step_kernels(ws, None, self._ahead.normal[slot], tiles=ws["slots"][slot])
            elif self.use_graph:
                key = (B, self.memory.cur_capacity if self._graph_rng else 0, id(self.actor), self._norm_key())
                if self._graph is None or self._graph_key != key:
                    with H.CAPTURE_LOCK:
                        self._capture(ws, key)
                if not self._graph_rng:
                    self._draws(ws)
                self._replay(ws, self._graph)
            else:
                self._draw_and_step(ws)
            self.update_count += 1   # under the lock: update() reads it together with the device loss ring (free-running threads)
        return self.sleep_time

    def _run_in_one_graph(self, ws, n):
        """Whether `n` steps from here are one whole run of draws-ahead steps that may replay as ONE hipGraph."""
        return (self.use_graph and self._run_graphs and self._ahead is not None and n == ws["K"] and n > 1 and self._ahead.valid in (0, n)
                and (self._ahead.valid == 0 or self._ahead.pos == 0) and self.memory.cur_capacity < (1 << 28)
                and (not self.dp or graph_collective_enabled(self.pg)))

    @torch.no_grad()
    def learn_many(self, n):
        """`n` consecutive gradient steps: exactly what n `learn()` calls do -- the same draws, tiles and launches in the same
        order on this learner's queue, bit for bit.  When they are one whole run of draws-ahead steps (the critic_sample_ratio
        steps between two `update()` calls of the fixed-ratio loop, scripts/train_pql.py) they replay as ONE hipGraph instead of
        one per step: every graph boundary costs the queue ~5 us of device time (tools/probes/multistep_graph_probe.py: 603.8 ->
        599.0 us per step) and the host a launch.  Anything else (a partial run, per-step draws, eager mode, eager data-parallel
        collectives) is the loop of `learn()` calls itself."""
        n = int(n)
        if self.actor is None or n <= 0:
            return self.sleep_time
        B = int(self.cfg.algo.batch_size)
        with self._lock, torch.cuda.device(self.device), self._on_stream():
            ws = self._workspace(B)
            self._check_ahead()
            if self._run_in_one_graph(ws, n):
                if self._ahead.valid == 0:
                    self._prefetch(ws)
                key = (B, 0, id(self.actor), self._norm_key())
                if self._graph_key != key:
                    self._slot_graphs, self._run_graph, self._graph, self._graph_post, self._graph_key = {}, None, None, None, key
                if self._run_graph is None:
                    with H.CAPTURE_LOCK:
                        self._capture_run(ws, key)
                for _ in range(n):
                    self._ahead.take()
                self._run_graph.replay()
                self.update_count += n
                return self.sleep_time
            if self._ahead is not None and self._ahead.valid == 0 and n < ws["K"] and self.memory.cur_capacity < (1 << 28):
                self._prefetch(ws, steps=n)   # a partial run: fetch what its n steps will use, not K steps' worth
        for _ in range(n):
            self.learn()
        return self.sleep_time

    def _capture_run(self, ws, key):
        """All K steps of a run (slot 0 .. K-1, in order) in one hipGraph; the tiles and draws `_prefetch` left are in place."""
        def run():
            for slot in range(ws["K"]):
                self._step_kernels(ws, None, self._ahead.normal[slot], tiles=ws["slots"][slot])
        snap = self._snapshot()
        s = torch.cuda.Stream(self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            run()   # warm-up outside capture, on a side stream as torch requires
        torch.cuda.current_stream(self.device).wait_stream(s)
        self._restore(snap)
        g = self._new_graph()
        if self.dp:   # (a run graph under data parallel exists only with captured collectives)
            DP.drain_pending_collectives(self.pg)
        with torch.cuda.graph(g, stream=self._capture_stream, capture_error_mode="thread_local"):
            run()
        self._restore(snap)
        self._run_graph, self._graph_key = g, key

    def _replay(self, ws, g):
        """g: the step's hipGraph, or (data parallel, collectives kept eager) the list of its pieces: one graph up to the
        gradient + ONE all-reduce, or one graph per bucket with that bucket's all-reduce issued behind it; then the optimiser's."""
        if isinstance(g, list):
            for k, piece in enumerate(g):
                piece.replay()
                self._reducer.issue(ws["bucket_views"][k])
            self._reducer.wait()
            self._graph_post.replay()
            return
        g.replay()
        if self._graph_post is not None:   # data parallel: the collective stays outside the graphs
            self._allreduce_grads(ws)
            self._graph_post.replay()

    @torch.no_grad()
    def prepare(self):
        """Build the workspace and capture the step's hipGraph now instead of inside the first `learn()` (capture runs one
        step and restores every tensor and the RNG state it touched, so this changes nothing observable)."""
        if self.actor is None:
            return
        with self._lock, torch.cuda.device(self.device), self._on_stream():
            ws = self._workspace(int(self.cfg.algo.batch_size))
            if self.use_graph and self._ahead is not None and 0 < self.memory.cur_capacity < (1 << 28):
                key = (ws["B"], 0, id(self.actor), self._norm_key())
                if self._graph_key != key:
                    self._slot_graphs, self._run_graph, self._graph, self._graph_post, self._graph_key = {}, None, None, None, key
                off = self.gen.get_offset()
                self._prefetch(ws)              # (the captures' warm-up runs need real tiles; nothing is consumed: the
                for slot in range(ws["K"]):     #  generator is put back and the tiles are dropped)
                    if slot not in self._slot_graphs:
                        with H.CAPTURE_LOCK:
                            self._capture(ws, key, slot)
                if self._run_graph is None and self._run_graphs and ws["K"] > 1 and (not self.dp or graph_collective_enabled(self.pg)):
                    with H.CAPTURE_LOCK:
                        self._capture_run(ws, key)
                self._drop_ahead()
                self.gen.set_offset(off)
            elif self.use_graph:
                key = (ws["B"], self.memory.cur_capacity if self._graph_rng else 0, id(self.actor), self._norm_key())
                if self._graph is None or self._graph_key != key:
                    with H.CAPTURE_LOCK:
                        self._capture(ws, key)

    def _inject(self, dst, src, home):
        st = torch.cuda.current_stream(self.device)
        lease = H.acquire(src, st, home) if src.is_cuda else None
        dst.copy_(src.reshape(dst.shape), non_blocking=src.is_cuda)
        H.release(lease, st)

    def _capture(self, ws, key, slot=None):
        """Capture the whole step into a hipGraph.  With `algo.graph_rng` the RNG draws are captured too; the graph then bakes
        in cur_capacity (the randint bound) and is re-captured while the ring is still filling.  slot: the step reads the
        draws / tiles `_prefetch` left in that slot (no RNG, no gather inside the graph); one graph per slot."""
        if slot is None:
            step = lambda **kw: self._draw_and_step(ws, **kw)   # noqa: E731
        else:
            def step(upto_backward=False, draw=None, part=None):
                self._step_kernels(ws, None, self._ahead.normal[slot], upto_backward, tiles=ws["slots"][slot], part=part)
        # warm-up outside capture (lazy hipFuncSetAttribute / allocator state), on a side stream as torch requires
        snap = self._snapshot()
        s = torch.cuda.Stream(self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            step()
        torch.cuda.current_stream(self.device).wait_stream(s)
        self._restore(snap)
        g, g_post = self._new_graph(), None
        # PQL_DP_GRAPH_COLLECTIVE=1 (opt-in, RCCL only, rehearsed with a 1-rank group only): capture the all-reduce inside
        # ONE graph instead of splitting the step around an eager collective
        if not self.dp or graph_collective_enabled(self.pg):
            if self.dp:
                DP.drain_pending_collectives(self.pg)   # (the warm-up's eager all-reduce must have left the watchdog's list)
            with torch.cuda.graph(g, stream=self._capture_stream, capture_error_mode="thread_local"):
                step(draw=self._graph_rng)
        else:   # two graphs around the RCCL all-reduce (kept eager: no collective is ever captured)
            if self._buckets is not None:   # ... or one per gradient bucket, each followed by its own collective
                g = [g] + [self._new_graph() for _ in self._buckets[1:]]
                for k, piece in enumerate(g):
                    with torch.cuda.graph(piece, stream=self._capture_stream, capture_error_mode="thread_local"):
                        step(part=k, draw=self._graph_rng and k == 0)
            else:
                with torch.cuda.graph(g, stream=self._capture_stream, capture_error_mode="thread_local"):
                    step(upto_backward=True, draw=self._graph_rng)
            if slot is None or self._graph_post is None:   # (the optimiser graph is the same for every slot)
                g_post = self._new_graph()
                with torch.cuda.graph(g_post, stream=self._capture_stream, capture_error_mode="thread_local"):
                    self._step_post(ws)
            else:
                g_post = self._graph_post
        self._restore(snap)  # capture does not execute, but keep state exactly as before
        if slot is None:
            self._graph, self._graph_post, self._graph_key = g, g_post, key
        else:
            self._slot_graphs[slot] = g
            self._graph_post, self._graph_key = g_post, key

    def _new_graph(self):
        g = torch.cuda.CUDAGraph()
        if self.gen is not None and self._graph_rng:    # a private generator takes part in capture only when registered with the graph
            g.register_generator_state(self.gen)
        return g

    def _snapshot(self):
        return [t.clone() for t in (self.critic.arena.data, self.critic_target.arena.data, self.opt.m, self.opt.v,
                                    self.opt.step, self.loss_ring)], \
            (self.gen.get_state() if self.gen is not None else torch.cuda.get_rng_state(self.device))

    def _restore(self, snap):
        tensors, rng = snap
        for dst, src in zip((self.critic.arena.data, self.critic_target.arena.data, self.opt.m, self.opt.v, self.opt.step,
                             self.loss_ring), tensors):
            dst.copy_(src)
        self.repack()
        if self.gen is not None:
            self.gen.set_state(rng)
        else:
            torch.cuda.set_rng_state(rng, self.device)

    # ------------------------------------------------------------------------------------------
    def loss_mean(self):
        """Exact mean of the last 5 losses (Tracker(5).mean(), zero-filled before 5 steps); synchronises."""
        with torch.cuda.device(self.device), self._on_stream():
            vals = self.loss_ring.tolist()
        m = LaggedLoss.mean_of(vals, self.update_count)
        self.loss_tracker = Tracker(LOSS_RING)
        for t in range(self.update_count - min(self.update_count, LOSS_RING), self.update_count):
            self.loss_tracker.update(vals[t % LOSS_RING])
        return m

    def set_actor(self, actor, home=None):
        """Adopt new policy weights into the resident replica: a fenced flat-arena copy on this learner's stream; from
        another GPU through the copy streams (peer copy over xGMI) -- the reference pickles the module through Ray."""
        if self.actor is None or self.actor.layout.dims != actor.layout.dims:
            st = torch.cuda.current_stream(self.device)
            with H.LOCK:
                lease = H.acquire(actor, st, home)
                self.actor = deepcopy(actor).to(self.device)
                H.release(lease, st)
            self.actor.requires_grad_(False)
            self.pk_actor = PackedWeights(self.actor.layout, self.device) if self._fused else None
        elif actor is not self.actor:
            adopt_arena(self.actor, actor, self.device, home)
        if self.pk_actor is not None:
            self.pk_actor.refresh(self.actor.arena.data)

    @torch.no_grad()
    def update(self, actor, trajectory, normalize_tuple, sleep_time):
        """pql_v_learner.py:117-122.  Everything is enqueued on this learner's stream behind event fences
        (pql_amd.utils.handoff): the stream waits for the producers of `actor`, `trajectory` and the statistics, the
        producers' buffers are released when the copies / the ring insert have read them, and the critic handed back
        is a snapshot (double-buffered) that later optimiser steps do not touch."""
        home = torch.cuda.current_stream(self.device)   # the caller's stream, before we switch to ours
        with self._lock, torch.cuda.device(self.device), self._on_stream():
            st = torch.cuda.current_stream(self.device)
            self.set_actor(actor, home)
            with H.LOCK:
                lease = H.acquire(trajectory, st, home)
                self.memory.add_to_buffer(trajectory)
                H.release(lease, st)
            self.normalize_tuple = resident_norm(self, normalize_tuple, home)
            self._drop_ahead()   # ring contents, the randint bound and the statistics changed: later steps sample afresh
            loss = self._lagged.poll(self.update_count)
            self.sleep_time = sleep_time
            return self._published(), loss, self.update_count


def asyn_v_learner(learner, cfg, stop_event=None, max_in_flight=2):
    """Free-running pump (reference: a Ray task looping forever, :136-141).  Run it in a thread."""
    pump(learner, stop_event, max_in_flight)
