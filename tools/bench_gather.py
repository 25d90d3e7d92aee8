#!/usr/bin/env python3
"""Micro-benchmark of the fused replay gather (k_replay_gather_fast / _obs / _fused) at BASELINE shapes for a sweep of the tuning
hooks of the flag word.  Two timings per point (HIP events around hipGraph replays, fresh random indices per launch):
  b2b  -- launches back to back into one set of output tiles (what rounds 2-3 tuned on; the tiles then live in the Infinity Cache)
  iso  -- ONE launch at a time behind a 384-MB eviction write, rotating output-tile sets (bench.isolated_launch_ms: what the
          schedule pays, and what rocprofv3's per-launch durations show)
GPU only.
    python tools/bench_gather.py [cfg2|cfg5|cfg4|cfg2x8|cfg5x8|cfg4x8|p2|p2x4] ... [--quick] [--auto]
A name may carry a ring size, `cfg5x8@150M` (rows; the 1-KiB records of cfg #5 x 150 M = 154 GB): the same launch over a ring that
takes most of the card.  --auto: only the learner's own flag word (no sweep)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import isolated_launch_ms  # noqa: E402
from pql_amd import _lib as L  # noqa: E402
from pql_amd.replay.simple_replay import RecordRing, ReplayBuffer  # noqa: E402

# (obs, act, rows per launch, ring rows): one batch per launch (the per-step path) and the launches that serve 8 V-steps at once;
# act = -1: the P-learner's obs-only ring (one batch, and the 4-batch launch of the schedule)
CFG = {"cfg2": (88, 16, 8192, 1_000_000), "cfg5": (108, 21, 32768, 5_000_000), "cfg4": (211, 20, 8192, 2_000_000),
       "cfg2x8": (88, 16, 8 * 8192, 1_000_000), "cfg5x8": (108, 21, 8 * 32768, 5_000_000), "cfg4x8": (211, 20, 8 * 8192, 2_000_000),
       "p2": (88, -1, 8192, 1_000_000), "p2x4": (88, -1, 4 * 8192, 1_000_000), "p4x4": (211, -1, 4 * 8192, 2_000_000),
       "p5x4": (108, -1, 4 * 32768, 5_000_000),
       # "...s": the obs ring gathered into ONE destination (the actor's input tile: what the P-learner does since round 4)
       "p2s": (88, -1, 8192, 1_000_000), "p2x4s": (88, -1, 4 * 8192, 1_000_000)}
SETS, N_ISO = 4, 16


def run(name, quick=False, iters=30, auto=False):
    name, _, rows = name.partition("@")
    iters = max(4, min(iters, (1 << 23) // CFG[name][2]))   # (bound the index tensor for the 8-batch launches)
    O, A, B, cap = CFG[name]
    if rows:
        cap = int(float(rows.rstrip("Mk")) * {"M": 1e6, "k": 1e3}.get(rows[-1], 1))
    dev = torch.device("cuda:0")
    obs_only = A < 0
    if obs_only:
        ring = RecordRing(cap, O, -1, dev)
    else:
        rb = ReplayBuffer(cap, (O,), A, dev)
        rb.cur_capacity, rb.if_full = cap, True
        ring = rb.ring
    if cap * ring.rec_ld * 4 < (8 << 30):
        ring.records.normal_()
    else:   # (values do not matter to the timing; a 100-GB normal_ is 25 G Philox draws)
        ring.records.fill_(0.25)
    ld_sa, ld_o = L.ld(O + max(A, 16)), L.ld(O)
    f = dict(dtype=torch.float32, device=dev)
    single = name.endswith("s")
    tiles = [dict(x_sa=None if single else torch.zeros((B, ld_sa), **f), xn_sa=None if obs_only else torch.zeros((B, ld_sa), **f),
                  x_obs=torch.zeros((B, ld_o), **f) if obs_only else None,
                  rew=None if obs_only else torch.zeros(B, **f), done=None if obs_only else torch.zeros(B, **f)) for _ in range(SETS)]
    mean = torch.randn(O, device=dev) * 0.1; var = torch.rand(O, device=dev) + 0.5
    idx = torch.randint(cap, (max(iters, N_ISO) + 2, B), device=dev)
    if obs_only:
        alg = B * (2 * O * 4 + 8)
        real = B * (ring.rec_ld * 4 + 8 + ((0 if name.endswith("s") else ld_sa) + ld_o) * 4)
    else:
        alg = B * ((2 * O + A) * 4 + 4 + 1 + 8 + (2 * O + A) * 4 + 4 + 4)
        real = B * (ring.rec_ld * 4 + 8 + 2 * ld_sa * 4 + 8)

    def one(i, s=0, norm=True, flags=1):
        t = tiles[s]
        L.check(L.lib.pqlk_replay_gather_fused(C.byref(ring.desc), L.ptr(idx[i]), B, L.ptr(mean) if norm else None,
                                               L.ptr(var) if norm else None, 1e-4, flags, L.ptr(t["x_sa"]), ld_sa, L.ptr(t["xn_sa"]),
                                               L.ptr(t["x_obs"]), ld_o, L.ptr(t["rew"]), L.ptr(t["done"]), L.stream(dev)))

    def b2b(norm=True, flags=1):
        one(0, 0, norm, flags)
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.graph(g, stream=side):
            for i in range(iters):
                one(2 + i, 0, norm, flags)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    def iso(norm=True, flags=1):
        return isolated_launch_ms(lambda i, s: one(i, s, norm, flags), dev, N_ISO, SETS) * 1e3

    print(f"== {name}: O={O} A={A} B={B} rec={ring.rec_ld * 4}B  algorithmic {alg / 1e6:.2f} MB, moved {real / 1e6:.2f} MB", flush=True)
    print(f"  ring {cap} rows = {cap * ring.rec_ld * 4 / 1e9:.1f} GB", flush=True)
    Rs = () if auto else (1, 2, 4) if obs_only else (2, 4)
    wpcs = (8, 16, 24, 32) if obs_only else (8, 12, 16, 24, 32)
    if quick:
        wpcs = (12, 16, 24) if not obs_only else (8, 16, 32)
    for R in Rs:
        for wpc in wpcs:
            for nopad in ((1,) if quick else (0, 1)):
                for nt in ((0,) if obs_only else (0, 1)):
                    flags = 1 | (2 if nopad else 0) | (4 if nt else 0) | (R << 8) | (wpc << 12)   # include/pqlk.h: PQLK_GATHER_*
                    us = min(b2b(flags=flags) for _ in range(3))
                    ui = iso(flags=flags)
                    print(f"  R={R} waves/CU={wpc} nopad={nopad} nt={nt}: b2b {us:6.2f} us ({alg / us / 1e6 / 8:.3f} of 8 TB/s)   "
                          f"iso {ui:6.2f} us ({alg / ui / 1e6 / 8:.3f})   moved {real / ui / 1e6:5.2f} TB/s iso", flush=True)
    print(f"  auto (learner flags 3): b2b {b2b(flags=3):6.2f} us   iso {iso(flags=3):6.2f} us ({alg / iso(flags=3) / 1e6 / 8:.3f})", flush=True)
    print(f"  auto, no normalisation: b2b {b2b(False, 3):6.2f} us   iso {iso(False, 3):6.2f} us", flush=True)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    for n in (args or ["cfg2", "cfg5"]):
        run(n, quick="--quick" in sys.argv, auto="--auto" in sys.argv)
        torch.cuda.empty_cache()
