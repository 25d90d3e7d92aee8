"""C51 categorical projection with the reference's signature (pql/utils/distl_util.py:4-20)."""
import torch

from pql_amd import _lib as L


@torch.no_grad()
def projection(next_dist, reward, done, gamma, v_min=-10, v_max=10, num_atoms=51, support=None, device="cuda:0"):
    """Project r + (1-d)*gamma*z onto the fixed support.  One wave per row, per-bin accumulation in atom
    order (deterministic, unlike the reference's atomic index_add_ on GPU)."""
    p = next_dist.to(torch.float32).contiguous()
    L.require_gpu(p, "next_dist")
    B = p.shape[0]
    if support is None:
        support = torch.linspace(v_min, v_max, num_atoms, device=p.device)
    z = support.to(p.device, torch.float32).contiguous()
    rew = reward.reshape(-1).to(p.device, torch.float32).contiguous()
    dn = done.reshape(-1).to(p.device, torch.float32).contiguous()
    out = torch.empty_like(p)
    with torch.cuda.device(p.device):
        L.check(L.lib.pqlk_c51_project(L.ptr(p), L.ptr(rew), L.ptr(dn), L.ptr(z), float(gamma), float(v_min),
                                       float(v_max), int(num_atoms), B, L.ptr(out), L.stream(p.device)))
    return out
