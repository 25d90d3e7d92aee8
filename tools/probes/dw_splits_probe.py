#!/usr/bin/env python3
"""How many batch splits should the dW products use?  The critic backward of cfg #2 (batch 8192, twin [104, 512, 512, 256, 1]) with
`splits` = argv[1] (default 16): HIP-event time of the whole call replayed from a hipGraph; run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel split.  Round 4: is the layer-1 dW product (24 us at 0.46 of the MFMA roof with
64 x 64 tiles x 16 splits = 134 MB through the LDS-DMA path for 2.15 GFLOP) better served by 128 x 128 tiles x 32 splits (67 MB)?
    python tools/probes/dw_splits_probe.py 32 [actor]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from pql_amd import _lib as L  # noqa: E402
from pql_amd.models.mlp import ArenaLayout, PackedWeights, mlp_forward_raw  # noqa: E402


def main():
    splits = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    actor = len(sys.argv) > 2 and sys.argv[2] == "actor"
    dev = torch.device("cuda:0")
    B, O, A = 8192, 88, 16
    lay = ArenaLayout([O, 512, 512, 256, A], 1) if actor else ArenaLayout([O + A, 512, 512, 256, 1], 2)
    g = torch.Generator(device=dev).manual_seed(1)
    arena = (torch.rand(lay.total, device=dev, generator=g) - 0.5) * 0.1
    x = torch.zeros((B, lay.ld_in), device=dev)
    x[:, : lay.dims[0]] = torch.randn((B, lay.dims[0]), device=dev, generator=g)
    pk = PackedWeights(lay, dev).refresh(arena)
    acts = mlp_forward_raw(lay, arena, x, L.ACT_NONE, packed=pk, stash_all=True)
    dy = torch.randn((lay.n_nets, B, lay.ld_out), device=dev, generator=g) * 1e-3
    dy[:, :, lay.dims[-1]:] = 0
    grads = torch.zeros(lay.total, device=dev)
    ws = torch.empty(lay.bwd_ws_floats(B, splits), device=dev)

    def one():
        L.check(L.lib.pqlk_mlp_backward(C.byref(lay.desc), L.ptr(arena), L.ptr(x), lay.ld_in, B, L.ptr(acts), L.ptr(dy), L.ptr(grads), splits,
                                        None, 0, 0, 0, None, 0, L.ptr(ws), ws.numel(), L.stream(dev)))
    for _ in range(3):
        one()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side):
        for _ in range(10):
            one()
    gr.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    print(f"{'actor' if actor else 'twin critic'} backward, splits {splits}: {sorted(ts)[2]:.1f} us per call (gradient checksum {float(grads.double().abs().sum()):.6e})")


if __name__ == "__main__":
    main()
