#!/usr/bin/env python3
"""Probe: what does a hipGraph boundary cost between two V-learner steps?  The K = 8 steps between two update() calls replayed as
eight per-slot graphs (the learner's form) against ONE graph holding all eight (same launches, same tiles).

    python tools/probes/multistep_graph_probe.py [bench.py flags]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    sys.argv = ["bench.py", "--no-cpu-baseline"] + sys.argv[1:]
    args = bench.parse()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    torch.manual_seed(42)
    cfg, env, actor, v, p = bench.build_system(args, 0, 1, dev, None)
    bench.prefill(actor, v, p, env, cfg, args, dev)
    v.learn()
    torch.cuda.synchronize()
    ws = v._ws
    K = ws["K"]
    v._prefetch(ws)   # real tiles in every slot

    def step(slot):
        v._step_kernels(ws, None, v._ahead.normal[slot], tiles=ws["slots"][slot])

    def capture(fn):
        snap = v._snapshot()
        s = torch.cuda.Stream(dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            fn()
        torch.cuda.current_stream(dev).wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            fn()
        v._restore(snap)
        return g

    per_slot = [capture(lambda k=k: step(k)) for k in range(K)]
    whole = capture(lambda: [step(k) for k in range(K)])

    def run_slots():
        for g in per_slot:
            g.replay()

    res = {"eight graphs": [], "one graph": []}
    for r in range(9):
        for name, fn in (("eight graphs", run_slots), ("one graph", whole.replay)):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); e1.synchronize()
            res[name].append(e0.elapsed_time(e1) / (10 * K) * 1e3)
    for name, xs in res.items():
        xs.sort()
        print(f"{name:13s}: median {xs[len(xs) // 2]:.1f} us / V step   min {xs[0]:.1f}")


if __name__ == "__main__":
    main()
