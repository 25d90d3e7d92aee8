#!/usr/bin/env python3
"""Micro-benchmark of the fused replay gather (k_replay_gather_fast / _fused) at BASELINE shapes: HIP-event time per launch
over a hipGraph of launches with fresh random indices, for a sweep of the tuning hooks.  GPU only.
    python tools/bench_gather.py [cfg2|cfg5|cfg4] ..."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pql_amd import _lib as L  # noqa: E402
from pql_amd.replay.simple_replay import ReplayBuffer  # noqa: E402

# (obs, act, rows per launch, ring rows): one batch per launch (the per-step path) and the launches that serve 8 V-steps at once
CFG = {"cfg2": (88, 16, 8192, 1_000_000), "cfg5": (108, 21, 32768, 5_000_000), "cfg4": (211, 20, 8192, 2_000_000),
       "cfg2x8": (88, 16, 8 * 8192, 1_000_000), "cfg5x8": (108, 21, 8 * 32768, 5_000_000), "cfg4x8": (211, 20, 8 * 8192, 2_000_000)}


def run(name, iters=30):
    iters = max(4, min(iters, (1 << 23) // CFG[name][2]))   # (bound the index tensor for the 8-batch launches)
    O, A, B, cap = CFG[name]
    dev = torch.device("cuda:0")
    rb = ReplayBuffer(cap, (O,), A, dev)
    rb.ring.records.normal_()
    rb.cur_capacity, rb.if_full = cap, True
    ld_sa, ld_o = L.ld(O + A), L.ld(O)
    x_sa = torch.zeros((B, ld_sa), device=dev); xn_sa = torch.zeros((B, ld_sa), device=dev)
    rew = torch.zeros(B, device=dev); done = torch.zeros(B, device=dev)
    mean = torch.randn(O, device=dev) * 0.1; var = torch.rand(O, device=dev) + 0.5
    idx = torch.randint(cap, (iters + 2, B), device=dev)
    alg = B * ((2 * O + A) * 4 + 4 + 1 + 8 + (2 * O + A) * 4 + 4 + 4)
    real = B * (rb.ring.rec_ld * 4 + 8 + 2 * ld_sa * 4 + 8)

    def one(i, norm=True, flags=1):
        L.check(L.lib.pqlk_replay_gather_fused(C.byref(rb.ring.desc), L.ptr(idx[i]), B, L.ptr(mean) if norm else None,
                                               L.ptr(var) if norm else None, 1e-4, flags, L.ptr(x_sa), ld_sa, L.ptr(xn_sa), None, ld_o,
                                               L.ptr(rew), L.ptr(done), L.stream(dev)))

    def timed(norm=True, flags=1):
        one(0, norm, flags)
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.graph(g, stream=side):
            for i in range(iters):
                one(2 + i, norm, flags)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    print(f"== {name}: O={O} A={A} B={B} rec={rb.ring.rec_ld * 4}B  algorithmic {alg / 1e6:.2f} MB, moved {real / 1e6:.2f} MB")
    for R in (2, 4):
        for wpc in (8, 12, 16, 24, 32):
            for nopad in (0, 1):
                for nt in (0, 1):
                    flags = 1 | (2 if nopad else 0) | (4 if nt else 0) | (R << 8) | (wpc << 12)   # include/pqlk.h: PQLK_GATHER_*
                    us = min(timed(flags=flags) for _ in range(3))
                    print(f"  R={R} waves/CU={wpc} nopad={nopad} nt={nt}: {us:6.2f} us  alg {alg / us / 1e6:5.2f} TB/s ({alg / us / 1e6 / 8:.3f} of 8)  moved {real / us / 1e6:5.2f} TB/s")
    print(f"  auto, no normalisation: {timed(False):6.2f} us")


if __name__ == "__main__":
    for n in (sys.argv[1:] or ["cfg2", "cfg5"]):
        run(n)
