import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pql_amd import _lib as L
from pql_amd.models.mlp import ArenaLayout, PackedWeights, mlp_forward_raw, output_view
dev = torch.device("cuda:0")
O, A, B = 88, 16, 8192
lay = ArenaLayout([O + A, 512, 512, 256, 1], 2)
g = torch.Generator(device=dev).manual_seed(1)
arena = (torch.rand(lay.total, device=dev, generator=g) - 0.5) * 0.1
x = torch.zeros((B, lay.ld_in), device=dev); x[:, :O + A] = torch.randn((B, O + A), device=dev, generator=g)
a_out = torch.zeros((1, B, L.ld(A)), device=dev); a_out[0, :, :A] = torch.tanh(torch.randn((B, A), device=dev, generator=g))
pk = PackedWeights(lay, dev).refresh(arena)
acts = mlp_forward_raw(lay, arena, x, L.ACT_NONE, packed=pk, stash_all=True)
q = output_view(lay, acts, B)
dy = torch.zeros((2, B, lay.ld_out), device=dev); ring = torch.zeros(5, device=dev); slot = torch.zeros(1, dtype=torch.int32, device=dev)
scratch = torch.zeros(2048, device=dev); owner = torch.zeros(B, dtype=torch.uint8, device=dev)
dz = torch.zeros((1, B, L.ld(A)), device=dev)
wsd = torch.empty(lay.bwd_ws_floats(B, 1), device=dev)
wsc = torch.empty(int(L.lib.pqlk_dpg_backward_ws_floats(C.byref(lay.desc), B)), device=dev)

def loss():
    L.check(L.lib.pqlk_dpg_loss_owner(L.ptr(q), lay.ld_out, 1, None, B, L.ptr(dy), L.ptr(ring), L.ptr(slot), 5, L.ptr(scratch), C.c_void_p(owner.data_ptr()), L.stream(dev)))
def dense():
    L.check(L.lib.pqlk_mlp_backward(C.byref(lay.desc), L.ptr(arena), L.ptr(x), lay.ld_in, B, L.ptr(acts), L.ptr(dy), None, 1, L.ptr(dz), L.ld(A), O, A, L.ptr(a_out), L.ld(A), L.ptr(wsd), wsd.numel(), L.stream(dev)))
def compact():
    L.check(L.lib.pqlk_dpg_critic_backward(C.byref(lay.desc), L.ptr(arena), L.ptr(x), lay.ld_in, B, L.ptr(acts), L.ptr(dy), L.ptr(dz), L.ld(A), O, A, L.ptr(a_out), L.ld(A), C.c_void_p(owner.data_ptr()), L.ptr(wsc), wsc.numel(), L.stream(dev)))
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for case in ("natural", "all_net1", "all_ties"):
    if case == "all_net1": q[1, :, 0] = q[0, :, 0] - 1
    if case == "all_ties": q[1, :, 0] = q[0, :, 0]
    loss(); torch.cuda.synchronize()
    print(case, "owner", torch.bincount(owner.long(), minlength=4).tolist(), "dense %.1f us" % timeit(dense), "compact %.1f us" % timeit(compact))
