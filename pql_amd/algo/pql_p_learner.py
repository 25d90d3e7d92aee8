"""P-learner: the policy side of Parallel Q-Learning on one MI355X.

Drop-in for `pql/algo/pql_p_learner.py`: `PQLPLearner(obs_dim, action_dim, cfg)` with `start()`, `learn()`,
`update(critic, obs, normalize_tuple, sleep_time)` and the pump `asyn_p_learner`.

learn()  (reference :47-64)  ->  one launch sequence
    randint -> fused obs gather+normalise -> actor fwd (tanh; action dropped into the critic input) ->
    frozen twin-critic fwd -> DPG loss -mean(min Q) + dL/dQ -> critic bwd, dX ONLY, both nets summed in
    one GEMM, chained through tanh' in the epilogue -> actor bwd (dW + dX) -> clip + AdamW.
`critic.requires_grad_(False)` of the reference (:54) is structural here: no dW GEMM is launched for the critic.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import threading
from copy import deepcopy

import torch

from pql_amd import _lib as L
from pql_amd.algo.pql_v_learner import (GATHER_FLAGS, LOSS_RING, LaggedLoss, _AdamState, _cfg_get, adopt_arena, allreduce_sum, apply_optimizer,
                                        apply_optimizer_fused, f32_recip, graph_collective_enabled, pump, resident_norm)
from pql_amd.models import model_name_to_path
from pql_amd.models.mlp import PackedWeights, default_splits, mlp_forward_raw, output_view
from pql_amd.replay.simple_replay import RecordRing, _obs_width, ring_plan
from pql_amd.utils import handoff as H
from pql_amd.utils.dp import drain_pending_collectives
from pql_amd.utils import rng as R
from pql_amd.utils.common import Tracker, load_class_from_path


class PQLPLearner:
    def __init__(self, obs_dim, action_dim, cfg, process_group=None):
        self.cfg = cfg
        self.obs_dim = obs_dim
        self.action_dim = int(action_dim)
        if not torch.cuda.is_available():
            raise L.PqlkError("PQLPLearner needs an MI355X (no CPU path)")
        self.device = torch.device(f"cuda:{int(cfg.algo.p_learner_gpu)}")
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        # dp: the collective is issued even for a 1-rank group, so the RCCL path can be rehearsed on one GPU
        self.dp = process_group is not None
        algo = cfg.algo
        act_class = load_class_from_path(algo.act_class, model_name_to_path[algo.act_class])
        hidden = _cfg_get(algo, "hidden_layers")
        hidden = list(hidden) if hidden is not None else None
        with torch.cuda.device(self.device):
            self.actor = act_class(self.obs_dim, self.action_dim, hidden_layers=hidden).to(self.device)
        if cfg.artifact is not None:
            raise NotImplementedError("W&B artifact download is out of scope (no network); load a local state_dict instead")
        self.opt = _AdamState(self.actor.arena.data)
        self._fused = bool(_cfg_get(algo, "fused", True))
        self._fold_loss = bool(_cfg_get(algo, "fused_tail", True))   # see PQLVLearner
        self._fused_tail = not self.dp and self._fold_loss
        self.pk_actor = PackedWeights(self.actor.layout, self.device) if self._fused else None
        self.pk_critic = None
        self.critic = None

        # obs-only replay (reference :32-37: a bare (memory_size, obs) tensor + inline pointer logic)
        self.memory_size = int(algo.memory_size)
        self.ring = RecordRing(self.memory_size, _obs_width(obs_dim), -1, self.device)
        self.next_p = 0
        self.if_full = False
        self.cur_capacity = 0

        self.loss_tracker = Tracker(LOSS_RING)
        self.loss_ring = torch.zeros(LOSS_RING, dtype=torch.float32, device=self.device)
        self._lagged = LaggedLoss(self.loss_ring)
        self.update_count = 0
        self.normalize_tuple = None
        self.sleep_time = 0.01
        self.use_graph = bool(_cfg_get(algo, "graph", False))
        self._graph_rng = bool(_cfg_get(algo, "graph_rng", False))   # False: the randint is issued in front of the graph
        self.stream = torch.cuda.Stream(self.device) if bool(_cfg_get(algo, "streams", False)) else None
        # start()/update() hand out double-buffered snapshots of the actor (a pickled copy in the reference)
        self._pub = H.ArenaPublisher(self.actor)
        self._lock = threading.RLock()   # learn() / update() are FIFO like calls on a Ray actor
        self._capture_stream = torch.cuda.Stream(self.device)   # torch's default capture stream is shared by every graph
        self.gen = torch.Generator(device=self.device)   # own generator: see PQLVLearner.__init__
        self.gen.manual_seed(int(torch.randint(0, 2 ** 62, (1,)).item()))
        # algo.rng / algo.prefetch_steps: see PQLVLearner.__init__ (here: one randint per step, P-steps of one rollout iteration)
        self._rng_mode = str(_cfg_get(algo, "rng", "auto"))
        if self._rng_mode != "torch":   # see PQLVLearner.__init__
            with torch.cuda.device(self.device):
                R.verified(self.device)
        ratio = max(1, int(_cfg_get(algo, "critic_sample_ratio", 8)) // max(1, int(_cfg_get(algo, "critic_actor_ratio", 2))))
        self._depth = max(1, int(_cfg_get(algo, "prefetch_steps_p", ratio)))
        self._ahead = None
        self._ws = None
        self._graph = None
        self._graph_post = None
        self._graph_key = None
        self._slot_graphs = {}
        self._run_graph = None   # all K draws-ahead steps of one run in ONE hipGraph (learn_many)
        self._run_graphs = bool(_cfg_get(cfg.algo, "run_graph", True))

    @property
    def memory(self):
        """(memory_size, obs_dim) view of the ring, the reference's attribute name (:34)."""
        return self.ring.records[:, : self.ring.O]

    def start(self):
        with self._lock, torch.cuda.device(self.device), self._on_stream():
            return self._published(), self.update_count, self.loss_tracker.mean()

    def _on_stream(self):
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def _published(self):
        """The actor as handed to other components: a snapshot taken on this learner's queue."""
        return self._pub.publish()

    def use_private_rng(self, seed):
        """Re-seed this learner's generator."""
        self.gen.manual_seed(int(seed))
        self._graph = None
        self._drop_ahead()

    def _drop_ahead(self):
        if self._ahead is not None:
            self._ahead.invalidate()

    def _data_stamp(self):
        """See PQLVLearner._data_stamp: ring contents, randint bound, identity of the statistics."""
        nt = self.normalize_tuple
        return (self.ring.version, self.cur_capacity, None if nt is None else (id(nt[0]), id(nt[1]), float(nt[2])))

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
        return "philox" if self._ahead is not None else "torch"

    def _want_ahead(self):
        if self._rng_mode == "torch" or self._graph_rng:
            return False
        ok = R.verified(self.device) is not None
        if not ok and self._rng_mode == "philox":
            raise L.PqlkError("algo.rng=philox: pqlk_philox_draws does not reproduce torch.randint on this device; use auto / torch")
        return ok

    def ready_to_learn(self):
        return self.critic is not None

    def fence(self):
        ev = torch.cuda.Event()
        ev.record(self.stream if self.stream is not None else torch.cuda.current_stream(self.device))
        return ev

    def synchronize(self):
        self.fence().synchronize()

    def _workspace(self, B):
        if self._ws is not None and self._ws["B"] == B:
            return self._ws
        f = dict(dtype=torch.float32, device=self.device)
        O, A = self.ring.O, self.action_dim
        al, cl = self.actor.layout, self.critic.layout
        # (the actor's input tile is 64-float aligned -- 128 wide for 88 observations: its layer-1 dW product then reads whole 64-column
        #  tiles of it and takes the LDS-DMA main loop like every other backward GEMM; the extra columns are never written and stay zero)
        ws = dict(B=B, ld_sa=L.ld(O + A), ld_o=(L.ld(O) + 63) // 64 * 64, ld_a=L.ld(A))
        want = self._want_ahead()   # draws + gathered tiles of the next K steps (see PQLVLearner._workspace)
        K = self._depth if want else 1
        self._ahead = R.DrawAhead(self.gen, self.device, B, None, K, R.verified(self.device)) if want else None
        self._slot_graphs, self._run_graph = {}, None
        ws["K"] = K
        # round 4: loss + partition + compact head in one launch, the actor's head backward inside the action-slice launch
        # (pqlk_dpg_backward_fused: 16 -> 13 launches per step); needs the loss fold of the optimiser launch and a fused critic forward
        K_atoms = int(getattr(self.critic, "num_atoms", 1))
        ws["dpg_fused"] = bool(_cfg_get(self.cfg.algo, "dpg_fused", True) and self._fold_loss and K_atoms == 1 and self.pk_critic is not None
                               and self.pk_critic.tensor is not None and self.pk_actor is not None and self.pk_actor.tensor is not None
                               and L.lib.pqlk_dpg_fused_ok(C.byref(cl.desc), C.byref(al.desc), B))
        # ... and then the critic reads its input where the two halves of torch.cat((obs, action)) already lie -- the actor's input
        # tile and the actor's output block (pqlk_mlp_forward_qc's second source) -- so the gather writes the observations ONCE
        # (no [obs | action] tile at all: 42 -> 24 MB moved per 4-step launch at cfg #2)
        ws["split_in"] = bool(ws["dpg_fused"] and O % 4 == 0 and _cfg_get(self.cfg.algo, "dpg_split_input", True))
        ws["x_sa_all"] = None if ws["split_in"] else torch.zeros((K, B, ws["ld_sa"]), **f)
        ws["x_obs_all"] = torch.zeros((K, B, ws["ld_o"]), **f)
        ws["slots"] = [dict(x_sa=None if ws["split_in"] else ws["x_sa_all"][k], x_obs=ws["x_obs_all"][k]) for k in range(K)]
        ws.update(ws["slots"][0])
        ws["idx"] = torch.zeros(B, dtype=torch.int64, device=self.device)
        ws["acts_a"] = torch.empty(al.acts_floats(B), **f)
        ws["acts_c"] = torch.empty(cl.acts_floats(B), **f)
        ws["dy_c"] = torch.zeros((2, B, cl.ld_out), **f)
        ws["dz_a"] = torch.zeros((1, B, ws["ld_a"]), **f)   # dL/d(actor pre-tanh); pad columns stay zero
        ws["grads"] = torch.zeros(al.total, **f)
        ws["splits"] = default_splits(B, _cfg_get(self.cfg.algo, "dw_splits", 16))
        ws["bwd_c"] = torch.empty(int(L.lib.pqlk_dpg_backward_ws_floats(C.byref(cl.desc), B)), **f)
        ws["owner"] = torch.zeros(B, dtype=torch.uint8, device=self.device)   # which net(s) own each sample's min(Q1, Q2)
        ws["bwd_a"] = torch.empty(al.bwd_ws_floats(B, ws["splits"]), **f)
        ws["scratch"] = torch.zeros(2048, **f)
        if ws["dpg_fused"]:
            ws["qc"] = torch.zeros((2, B), **f)
            ws["head_parts"] = int(L.lib.pqlk_dpg_fused_head_parts(B))
            ws["mn_ptr"] = C.c_void_p(ws["bwd_c"].data_ptr() + 4 * int(L.lib.pqlk_dpg_fused_mn_offset(C.byref(cl.desc), B)))
            ws["loss_parts"] = int(L.lib.pqlk_dpg_fused_loss_parts())
        self._ws = ws
        self.repack()
        return ws

    def repack(self):
        if self.pk_actor is not None:
            self.pk_actor.refresh(self.actor.arena.data)
        if self.pk_critic is not None:
            self.pk_critic.refresh(self.critic.arena.data)

    def _gather(self, ws, idx, rows, x_sa, x_obs):
        mean, var, eps = (None, None, 0.0)
        if self.cfg.algo.obs_norm and self.normalize_tuple is not None:
            mean, var, eps = self.normalize_tuple
        L.check(L.lib.pqlk_replay_gather_fused(C.byref(self.ring.desc), L.ptr(idx), rows, L.ptr(mean), L.ptr(var), float(eps), GATHER_FLAGS,
                                               L.ptr(x_sa), ws["ld_sa"], None, L.ptr(x_obs), ws["ld_o"], None, None,
                                               L.stream(self.device)))

    def _prefetch(self, ws, steps=None):
        """The next K steps' indices (torch's numbers, one launch) and ONE gather of their K x B observation rows (`steps` < K: of
        that many steps only, see PQLVLearner._prefetch)."""
        Kp = ws["K"] if steps is None else max(1, min(ws["K"], int(steps)))
        self._ahead.refill(self.cur_capacity, Kp)
        self._gather(ws, self._ahead.idx, Kp * ws["B"], ws["x_sa_all"], ws["x_obs_all"])
        self._ahead_stamp = self._data_stamp()

    def _step_kernels(self, ws, idx, upto_backward=False, tiles=None):
        algo, dev, B = self.cfg.algo, self.device, ws["B"]
        O, A = self.ring.O, self.action_dim
        st = L.stream(dev)
        if tiles is None:   # per-step path: gather `idx` into slot 0 here
            tiles = ws["slots"][0]
            self._gather(ws, idx, B, tiles["x_sa"], tiles["x_obs"])
        ws = dict(ws, **tiles)
        al, cl = self.actor.layout, self.critic.layout
        x_act = None if ws["split_in"] else ws["x_sa"][:, O:]   # (split input: the critic reads the action out of the actor's output block)
        mlp_forward_raw(al, self.actor.arena.data, ws["x_obs"], L.ACT_TANH, acts=ws["acts_a"], out2=x_act, packed=self.pk_actor,
                        stash_all=True)
        tail = self._fused_tail   # see PQLVLearner._step_kernels
        a_out = output_view(al, ws["acts_a"], B)  # (1, B, ld_a): tanh output, for the tanh' chain
        if ws["dpg_fused"]:
            if ws["split_in"]:
                L.check(L.lib.pqlk_mlp_forward_qc(C.byref(cl.desc), L.ptr(self.critic.arena.data), L.ptr(self.pk_critic.tensor), 1, L.ptr(ws["x_obs"]),
                                                  ws["ld_o"], L.ptr(a_out), ws["ld_a"], O, B, L.ptr(ws["acts_c"]), L.ptr(ws["qc"]), st))
            else:
                L.check(L.lib.pqlk_mlp_forward_qc(C.byref(cl.desc), L.ptr(self.critic.arena.data), L.ptr(self.pk_critic.tensor), 1, L.ptr(ws["x_sa"]),
                                                  ws["ld_sa"], None, 0, 0, B, L.ptr(ws["acts_c"]), L.ptr(ws["qc"]), st))
            L.check(L.lib.pqlk_dpg_backward_fused(C.byref(cl.desc), L.ptr(self.critic.arena.data), L.ptr(ws["x_sa"]), ws["ld_sa"], B,
                                                  L.ptr(ws["acts_c"]), L.ptr(ws["qc"]), L.ptr(ws["dz_a"]), ws["ld_a"], O, L.ptr(a_out), ws["ld_a"],
                                                  L.ptr(ws["scratch"]), L.ptr(ws["bwd_c"]), ws["bwd_c"].numel(), C.byref(al.desc),
                                                  L.ptr(self.actor.arena.data), L.ptr(ws["acts_a"]), L.ptr(ws["bwd_a"]), ws["bwd_a"].numel(),
                                                  ws["splits"], st))
            L.check(L.lib.pqlk_mlp_backward_tail(C.byref(al.desc), L.ptr(self.actor.arena.data), L.ptr(ws["x_obs"]), ws["ld_o"], B,
                                                 L.ptr(ws["acts_a"]), L.ptr(ws["grads"]), ws["splits"], L.ptr(ws["bwd_a"]), ws["bwd_a"].numel(),
                                                 L.ptr(self.opt.scratch) if tail else None, L.ptr(self.opt.step) if tail else None,
                                                 ws["head_parts"], ws["mn_ptr"], st))
            if upto_backward:
                return
            self._allreduce_grads(ws)
            self._step_post(ws)
            return
        mlp_forward_raw(cl, self.critic.arena.data, ws["x_sa"], L.ACT_NONE, acts=ws["acts_c"], packed=self.pk_critic,
                        stash_all=True)   # the dX chain through the frozen critic needs its activations (ELU')
        q = output_view(cl, ws["acts_c"], B)
        K = int(getattr(self.critic, "num_atoms", 1))
        z = getattr(self.critic, "z_atoms", None) if K > 1 else None
        L.check(L.lib.pqlk_dpg_loss_owner(L.ptr(q), cl.ld_out, K, L.ptr(z), B, L.ptr(ws["dy_c"]), None if self._fold_loss else L.ptr(self.loss_ring),
                                          L.ptr(self.opt.step), LOSS_RING, L.ptr(ws["scratch"]), C.c_void_p(ws["owner"].data_ptr()), st))
        # dX-only chain through the frozen critic; with scalar Q heads it runs over the samples partitioned by the net that
        # attained min(Q1, Q2): the other net's rows of every dZ are exactly zero (csrc/minnet.h)
        L.check(L.lib.pqlk_dpg_critic_backward(C.byref(cl.desc), L.ptr(self.critic.arena.data), L.ptr(ws["x_sa"]), ws["ld_sa"], B,
                                               L.ptr(ws["acts_c"]), L.ptr(ws["dy_c"]), L.ptr(ws["dz_a"]), ws["ld_a"], O, A,
                                               L.ptr(a_out), ws["ld_a"], C.c_void_p(ws["owner"].data_ptr()) if K == 1 else None,
                                               L.ptr(ws["bwd_c"]), ws["bwd_c"].numel(), st))
        if tail:
            L.check(L.lib.pqlk_mlp_backward_norm(C.byref(al.desc), L.ptr(self.actor.arena.data), L.ptr(ws["x_obs"]), ws["ld_o"], B,
                                                 L.ptr(ws["acts_a"]), L.ptr(ws["dz_a"]), L.ptr(ws["grads"]), ws["splits"], None, 0, 0,
                                                 0, None, 0, L.ptr(ws["bwd_a"]), ws["bwd_a"].numel(), L.ptr(self.opt.scratch),
                                                 L.ptr(self.opt.step), st))
        else:
            L.check(L.lib.pqlk_mlp_backward(C.byref(al.desc), L.ptr(self.actor.arena.data), L.ptr(ws["x_obs"]), ws["ld_o"], B,
                                            L.ptr(ws["acts_a"]), L.ptr(ws["dz_a"]), L.ptr(ws["grads"]), ws["splits"], None, 0, 0, 0,
                                            None, 0, L.ptr(ws["bwd_a"]), ws["bwd_a"].numel(), st))
        if upto_backward:
            return
        self._allreduce_grads(ws)
        self._step_post(ws)

    def _allreduce_grads(self, ws):
        if self.dp:
            allreduce_sum(ws["grads"], self.pg)

    def _step_post(self, ws):
        algo = self.cfg.algo
        if self._fold_loss:
            K = int(getattr(self.critic, "num_atoms", 1))
            apply_optimizer_fused(self.actor.layout, self.actor.arena.data, ws["grads"], self.opt, None, algo.actor_lr,
                                  algo.max_grad_norm, 0.0, self.pk_actor, None, ws["scratch"],
                                  ws["loss_parts"] if ws["dpg_fused"] else L.lib.pqlk_loss_parts(ws["B"], K),
                                  f32_recip(ws["B"], sign=-1.0), self.loss_ring, self.device, norm_in_backward=self._fused_tail,
                                  grad_scale=1.0 / self.world)
            return
        apply_optimizer(self.actor.arena.data, ws["grads"], self.opt, None, algo.actor_lr, algo.max_grad_norm, 0.0,
                        1.0 / self.world, self.device, layout=self.actor.layout, packed=self.pk_actor)

    def _draws(self, ws):
        torch.randint(self.cur_capacity, (ws["B"],), generator=self.gen, out=ws["idx"])  # the only draw (:49), no copy launch

    def _draw_and_step(self, ws, upto_backward=False, draw=True):
        if draw:
            self._draws(ws)
        self._step_kernels(ws, ws["idx"], upto_backward)

    @torch.no_grad()
    def learn(self, indices=None):
        if self.critic is None:
            return self.sleep_time
        B = int(self.cfg.algo.batch_size)
        home = torch.cuda.current_stream(self.device)
        with self._lock, torch.cuda.device(self.device), self._on_stream():
            ws = self._workspace(B)
            if indices is not None:   # injected draw: arrives on the caller's stream (or from the host)
                st = torch.cuda.current_stream(self.device)
                lease = H.acquire(indices, st, home) if indices.is_cuda else None
                ws["idx"].copy_(indices.reshape(-1), non_blocking=indices.is_cuda)
                H.release(lease, st)
                self._step_kernels(ws, ws["idx"])
            elif self._ahead is not None and self.cur_capacity < (1 << 28):   # see PQLVLearner.learn
                self._check_ahead()
                if self._ahead.valid == 0:
                    self._prefetch(ws)
                slot = self._ahead.take()
                if self.use_graph:
                    key = (B, 0, id(self.critic), self._norm_key())
                    if self._graph_key != key:
                        self._slot_graphs, self._run_graph, self._graph, self._graph_post, self._graph_key = {}, None, None, None, key
                    if slot not in self._slot_graphs:
                        with H.CAPTURE_LOCK:
                            self._capture(ws, key, slot)
                    self._slot_graphs[slot].replay()
                    if self._graph_post is not None:
                        self._allreduce_grads(ws)
                        self._graph_post.replay()
                else:
                    self._step_kernels(ws, None, tiles=ws["slots"][slot])
            elif self.use_graph:
                key = (B, self.cur_capacity if self._graph_rng else 0, id(self.critic), self._norm_key())
                if self._graph is None or self._graph_key != key:
                    with H.CAPTURE_LOCK:
                        self._capture(ws, key)
                if not self._graph_rng:   # draws in front of the graph (see PQLVLearner.__init__)
                    self._draws(ws)
                self._graph.replay()
                if self._graph_post is not None:
                    self._allreduce_grads(ws)
                    self._graph_post.replay()
            else:
                self._draw_and_step(ws)
            self.update_count += 1   # under the lock (see PQLVLearner.learn)
        return self.sleep_time

    @torch.no_grad()
    def learn_many(self, n):
        """`n` consecutive actor steps, exactly what n `learn()` calls do; one whole run of draws-ahead steps replays as ONE
        hipGraph (see PQLVLearner.learn_many)."""
        n = int(n)
        if self.critic is None or n <= 0:
            return self.sleep_time
        B = int(self.cfg.algo.batch_size)
        with self._lock, torch.cuda.device(self.device), self._on_stream():
            ws = self._workspace(B)
            self._check_ahead()
            if (self.use_graph and self._run_graphs and self._ahead is not None and n == ws["K"] and n > 1 and self._ahead.valid in (0, n)
                    and (self._ahead.valid == 0 or self._ahead.pos == 0) and self.cur_capacity < (1 << 28)
                    and (not self.dp or graph_collective_enabled(self.pg))):
                if self._ahead.valid == 0:
                    self._prefetch(ws)
                key = (B, 0, id(self.critic), self._norm_key())
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
            if self._ahead is not None and self._ahead.valid == 0 and n < ws["K"] and self.cur_capacity < (1 << 28):
                self._prefetch(ws, steps=n)   # a partial run: fetch what its n steps will use
        for _ in range(n):
            self.learn()
        return self.sleep_time

    def _capture_run(self, ws, key):
        """All K steps of a run (slot 0 .. K-1, in order) in one hipGraph; the tiles `_prefetch` left are in place."""
        def run():
            for slot in range(ws["K"]):
                self._step_kernels(ws, None, tiles=ws["slots"][slot])
        snap = [t.clone() for t in self._state()]
        rng = self._rng_state()
        s = torch.cuda.Stream(self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            run()   # warm-up outside capture, on a side stream as torch requires
        torch.cuda.current_stream(self.device).wait_stream(s)
        for dst, src in zip(self._state(), snap):
            dst.copy_(src)
        self.repack()
        g = self._new_graph()
        if self.dp:   # (a run graph under data parallel exists only with captured collectives)
            drain_pending_collectives(self.pg)
        with torch.cuda.graph(g, stream=self._capture_stream, capture_error_mode="thread_local"):
            run()
        for dst, src in zip(self._state(), snap):
            dst.copy_(src)
        self.repack()
        self._set_rng_state(rng)
        self._run_graph, self._graph_key = g, key

    @torch.no_grad()
    def prepare(self):
        """Workspace + hipGraph capture now instead of inside the first `learn()` (side-effect free, see PQLVLearner.prepare)."""
        if self.critic is None:
            return
        with self._lock, torch.cuda.device(self.device), self._on_stream():
            ws = self._workspace(int(self.cfg.algo.batch_size))
            if self.use_graph and self._ahead is not None and 0 < self.cur_capacity < (1 << 28):
                key = (ws["B"], 0, id(self.critic), self._norm_key())
                if self._graph_key != key:
                    self._slot_graphs, self._run_graph, self._graph, self._graph_post, self._graph_key = {}, None, None, None, key
                off = self.gen.get_offset()
                self._prefetch(ws)              # real tiles for the captures' warm-up runs; nothing is consumed
                for slot in range(ws["K"]):
                    if slot not in self._slot_graphs:
                        with H.CAPTURE_LOCK:
                            self._capture(ws, key, slot)
                if self._run_graph is None and self._run_graphs and ws["K"] > 1 and (not self.dp or graph_collective_enabled(self.pg)):
                    with H.CAPTURE_LOCK:
                        self._capture_run(ws, key)
                self._drop_ahead()
                self.gen.set_offset(off)
            elif self.use_graph:
                key = (ws["B"], self.cur_capacity if self._graph_rng else 0, id(self.critic), self._norm_key())
                if self._graph is None or self._graph_key != key:
                    with H.CAPTURE_LOCK:
                        self._capture(ws, key)

    def _state(self):
        return (self.actor.arena.data, self.opt.m, self.opt.v, self.opt.step, self.loss_ring)

    def _new_graph(self):
        g = torch.cuda.CUDAGraph()
        if self.gen is not None and self._graph_rng:
            g.register_generator_state(self.gen)
        return g

    def _rng_state(self):
        return self.gen.get_state() if self.gen is not None else torch.cuda.get_rng_state(self.device)

    def _set_rng_state(self, state):
        if self.gen is not None:
            self.gen.set_state(state)
        else:
            torch.cuda.set_rng_state(state, self.device)

    def _capture(self, ws, key, slot=None):
        if slot is None:
            step = lambda **kw: self._draw_and_step(ws, **kw)   # noqa: E731
        else:   # the step reads the tiles `_prefetch` left in that slot: no RNG, no gather inside the graph
            def step(upto_backward=False, draw=None):
                self._step_kernels(ws, None, upto_backward, tiles=ws["slots"][slot])
        snap = [t.clone() for t in self._state()]
        rng = self._rng_state()
        s = torch.cuda.Stream(self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            step()
        torch.cuda.current_stream(self.device).wait_stream(s)
        for dst, src in zip(self._state(), snap):
            dst.copy_(src)
        self.repack()
        self._set_rng_state(rng)
        g, g_post = self._new_graph(), None
        # PQL_DP_GRAPH_COLLECTIVE=1 (opt-in, RCCL only, rehearsed with a 1-rank group only): capture the all-reduce inside
        # ONE graph instead of splitting the step around an eager collective
        if not self.dp or graph_collective_enabled(self.pg):
            if self.dp:
                drain_pending_collectives(self.pg)   # (the warm-up's eager all-reduce must have left the watchdog's list)
            with torch.cuda.graph(g, stream=self._capture_stream, capture_error_mode="thread_local"):
                step(draw=self._graph_rng)
        else:
            with torch.cuda.graph(g, stream=self._capture_stream, capture_error_mode="thread_local"):
                step(upto_backward=True, draw=self._graph_rng)
            if slot is None or self._graph_post is None:   # (the optimiser graph is the same for every slot)
                g_post = self._new_graph()
                with torch.cuda.graph(g_post, stream=self._capture_stream, capture_error_mode="thread_local"):
                    self._step_post(ws)
            else:
                g_post = self._graph_post
        self._set_rng_state(rng)
        if slot is None:
            self._graph, self._graph_post, self._graph_key = g, g_post, key
        else:
            self._slot_graphs[slot] = g
            self._graph_post, self._graph_key = g_post, key

    def loss_mean(self):
        """Exact mean of the last 5 losses (Tracker(5).mean(), zero-filled before 5 steps); synchronises."""
        with torch.cuda.device(self.device), self._on_stream():
            vals = self.loss_ring.tolist()
        m = LaggedLoss.mean_of(vals, self.update_count)
        self.loss_tracker = Tracker(LOSS_RING)
        for t in range(self.update_count - min(self.update_count, LOSS_RING), self.update_count):
            self.loss_tracker.update(vals[t % LOSS_RING])
        return m

    def set_critic(self, critic, home=None):
        """Adopt new critic weights into a resident replica: fenced flat-arena copy on this learner's stream (through
        the copy streams / xGMI when the V-learner lives on another GPU) -- the reference pickles the whole module."""
        if self.critic is None or self.critic.layout.dims != critic.layout.dims:
            st = torch.cuda.current_stream(self.device)
            with H.LOCK:
                lease = H.acquire(critic, st, home)
                self.critic = deepcopy(critic).to(self.device)
                H.release(lease, st)
            self.critic.requires_grad_(False)
            if hasattr(self.critic, "z_atoms"):
                self.critic.z_atoms = self.critic.z_atoms.to(self.device)
                self.critic.device = self.device
            self.pk_critic = PackedWeights(self.critic.layout, self.device) if self._fused else None
        elif critic is not self.critic:
            adopt_arena(self.critic, critic, self.device, home)
        if self.pk_critic is not None:
            self.pk_critic.refresh(self.critic.arena.data)

    @torch.no_grad()
    def update(self, critic, obs, normalize_tuple, sleep_time):
        """pql_p_learner.py:66-85, enqueued on this learner's stream behind event fences (see PQLVLearner.update)."""
        home = torch.cuda.current_stream(self.device)
        with self._lock, torch.cuda.device(self.device), self._on_stream():
            st = torch.cuda.current_stream(self.device)
            self.set_critic(critic, home)
            self.sleep_time = sleep_time
            self.normalize_tuple = resident_norm(self, normalize_tuple, home)
            with H.LOCK:
                lease = H.acquire(obs, st, home)
                obs = obs.reshape(-1, self.ring.O).to(self.device, torch.float32).contiguous()
                self.add_capacity = obs.shape[0]
                segs, self.next_p, self.if_full, self.cur_capacity = ring_plan(self.next_p, self.if_full, self.memory_size,
                                                                               obs.shape[0])
                self.ring.insert_segments(segs, obs)
                H.release(lease, st)
            self._drop_ahead()   # ring contents, the randint bound and the statistics changed: later steps sample afresh
            loss = self._lagged.poll(self.update_count)
            return self._published(), loss, self.update_count


def asyn_p_learner(learner, cfg, stop_event=None, max_in_flight=2):
    """Free-running pump (reference: a Ray task, :99-104).  Run it in a thread."""
    pump(learner, stop_event, max_in_flight)
