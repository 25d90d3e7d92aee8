#!/usr/bin/env python3
"""Interleaved A/B timing of the V-learner's launches on one MI355X (HIP events over hipGraphs of repeated launches).

    python tools/kbench.py [--variants "base;td=0;rng=torch"] [--rounds 5] [--sections actor,target,critic,bwd,opt,vstep,pstep] [bench.py flags ...]

A variant is a comma-separated list of learner settings (`td=0|1`: TD loss inside the head backward or as its own launch;
`rng=auto|torch`: draws ahead + batched gather or per-step ATen draws; `env:NAME=value`: an environment switch the library reads
per call).  Every section is timed for every variant in every round (cdna_hip_programming.md rule 24: deltas come from
interleaved rounds in ONE process); the table prints median and min per (section, variant).  Kernel-level experiments of a tuning
round are built as a second variant behind a temporary switch and compared here; the switch goes once a variant is chosen
(round 3: DESIGN.md section 11).  GPU only."""
import argparse
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from pql_amd import _lib as L  # noqa: E402
from pql_amd.models.mlp import mlp_forward_raw  # noqa: E402


def graph_time(fn, dev, iters=20, reps=3):
    """us per call of `fn`, replayed `iters` times from one hipGraph (min over `reps` replays)."""
    for _ in range(2):
        fn()
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream(dev).wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(iters):
            fn()
    g.replay()
    best = 1e30
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="base")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--sections", default="actor,target,critic,bwd,opt,vstep,pstep")
    ap.add_argument("--steps", type=int, default=200)
    ns, rest = ap.parse_known_args()
    sys.argv = ["bench.py", "--no-cpu-baseline"] + rest
    args = bench.parse()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    torch.manual_seed(42)
    cfg, env, actor, v, p = bench.build_system(args, 0, 1, dev, None)
    bench.prefill(actor, v, p, env, cfg, args, dev)
    B = int(cfg.algo.batch_size)
    ws = v._workspace(B)
    al, cl = v.actor.layout, v.critic.layout
    O = v.memory.ring.O
    st = lambda: L.stream(dev)  # noqa: E731
    v.learn(); p.learn()
    torch.cuda.synchronize()

    def apply(variant):
        v._td_in_head = True
        v._rng_mode = p._rng_mode = "auto"
        for k in [k for k in os.environ if k.startswith("PQLK_AB_")]:
            del os.environ[k]
        if variant != "base":
            for kv in variant.split(","):
                k, val = kv.split("=")
                if k == "td":
                    v._td_in_head = bool(int(val))
                elif k == "rng":
                    v._rng_mode = p._rng_mode = val
                elif k.startswith("env:"):
                    os.environ["PQLK_AB_" + k[4:]] = val
                else:
                    raise SystemExit(f"unknown variant key {k!r}")
        v._ws = None
        v._graph = None
        p._ws = None
        p._graph = None
        p._workspace(B)
        return v._workspace(B)

    def sec_actor():
        mlp_forward_raw(al, v.actor.arena.data, ws["xn_sa"], L.ACT_TANH_NOISE, ws["draw"], 0.8, 0.2, ws["acts_a"], ws["xn_sa"][:, O:],
                        packed=v.pk_actor, stash_all=False)

    def sec_target():
        mlp_forward_raw(cl, v.critic_target.arena.data, ws["xn_sa"], L.ACT_NONE, acts=ws["acts_t"], packed=v.pk_target, stash_all=False)

    def sec_critic():
        mlp_forward_raw(cl, v.critic.arena.data, ws["x_sa"], L.ACT_NONE, acts=ws["acts_c"], packed=v.pk_critic, stash_all=True)

    def sec_bwd():
        L.check(L.lib.pqlk_mlp_backward(C.byref(cl.desc), L.ptr(v.critic.arena.data), L.ptr(ws["x_sa"]), ws["ld_sa"], B,
                                        L.ptr(ws["acts_c"]), L.ptr(ws["dy"]), L.ptr(ws["grads"]), ws["splits"], None, 0, 0, 0,
                                        None, 0, L.ptr(ws["bwd"]), ws["bwd"].numel(), st()))

    scratch_arena = [t.clone() for t in (v.critic.arena.data, v.opt.m, v.opt.v, v.critic_target.arena.data)]
    from pql_amd.algo.pql_v_learner import apply_optimizer

    def sec_opt():   # clip + AdamW + Polyak + re-pack on scratch copies (norm pass included: two launches)
        apply_optimizer(scratch_arena[0], ws["grads"], v.opt, scratch_arena[3], 5e-4, 0.5, 0.05, 1.0, dev, layout=cl,
                        packed=v.pk_critic, packed_target=v.pk_target)

    def sec_opt_nopack():   # the same without the fragment-ordered re-pack (what the scattered 16-B pack stores cost)
        apply_optimizer(scratch_arena[0], ws["grads"], v.opt, scratch_arena[3], 5e-4, 0.5, 0.05, 1.0, dev)

    def rate(fn, n):   # free-running learner steps (own stream, hipGraph replay): wall clock around a full device sync
        import time
        for _ in range(8):
            fn()
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / n * 1e6)
        return best

    sections = {"actor": lambda: graph_time(sec_actor, dev), "target": lambda: graph_time(sec_target, dev),
                "critic": lambda: graph_time(sec_critic, dev), "bwd": lambda: graph_time(sec_bwd, dev),
                "opt": lambda: graph_time(sec_opt, dev), "opt_nopack": lambda: graph_time(sec_opt_nopack, dev), "vstep": lambda: rate(v.learn, ns.steps),
                "pstep": lambda: rate(p.learn, ns.steps)}
    want = [s for s in ns.sections.split(",") if s]
    variants = ns.variants.split(";")
    res = {(s, va): [] for s in want for va in variants}
    for r in range(ns.rounds):
        for va in variants:
            ws = apply(va)
            v.learn(); p.learn()
            torch.cuda.synchronize()
            for s in want:
                res[(s, va)].append(sections[s]())
        print(f"round {r} done", file=sys.stderr, flush=True)
    apply("base")
    print(f"{'section':10s} " + " ".join(f"{va:>22s}" for va in variants) + "   (us: median / min)")
    for s in want:
        print(f"{s:10s} " + " ".join(f"{statistics.median(res[(s, va)]):10.1f} /{min(res[(s, va)]):8.1f}  " for va in variants))


if __name__ == "__main__":
    main()
