import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import detdata as dd
from pql_amd import _lib as L
from pql_amd.replay.simple_replay import ReplayBuffer
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
dev = torch.device("cuda:0")
O, A, cap, B = 8, 2, 500, 77
rb = ReplayBuffer(cap, (O,), A, device=dev)
data = (T(dd.uniform((cap, O), 1, -30, 30)), T(dd.uniform((cap, A), 2)), T(dd.uniform((cap, 1), 3)),
        T(dd.uniform((cap, O), 4, -30, 30)), T(dd.bernoulli((cap, 1), 5, 0.3)))
rb.add_to_buffer(tuple(t.to(dev) for t in data))
idx = T(dd.integers((B,), 6, cap)); mean, var = T(dd.uniform((O,), 7)), T(dd.uniform((O,), 8, 0.01, 4.0))
o = data[0][idx]
cpu = ((o - mean) / torch.sqrt(var + 1e-4)).clamp(-5, 5)
gpu_t = ((o.to(dev) - mean.to(dev)) / torch.sqrt(var.to(dev) + 1e-4)).clamp(-5, 5).cpu()
ld_sa = L.ld(O + A)
x_sa = torch.zeros((B, ld_sa), device=dev)
idx_d, mean_d, var_d = idx.to(dev), mean.to(dev), var.to(dev)
L.check(L.lib.pqlk_replay_gather_fused(C.byref(rb.ring.desc), L.ptr(idx_d), B, L.ptr(mean_d), L.ptr(var_d), 1e-4, 1,
                                       L.ptr(x_sa), ld_sa, None, None, 0, None, None, L.stream(dev)))
mine = x_sa[:, :O].cpu()
print("cpu vs torch-gpu mismatches", (cpu != gpu_t).sum().item(), "cpu vs mine", (cpu != mine).sum().item(), "torch-gpu vs mine", (gpu_t != mine).sum().item())
bad = (cpu != mine).nonzero()
for r, c in bad[:5].tolist():
    print(r, c, o[r, c].item(), mean[c].item(), var[c].item(), cpu[r, c].item(), mine[r, c].item(), gpu_t[r, c].item())
# stepwise on cpu
s = torch.sqrt(var + 1e-4); print("sqrt cpu vs gpu", (s != torch.sqrt(var.to(dev) + 1e-4).cpu()).sum().item())
d = (o - mean); print("sub", (d != (o.to(dev) - mean.to(dev)).cpu()).sum().item())
q = d / s; print("div cpu vs gpu", (q != (d.to(dev) / s.to(dev)).cpu()).sum().item())
q64 = (d.double() / s.double()).float(); print("cpu div vs f64-rounded", (q != q64).sum().item())
