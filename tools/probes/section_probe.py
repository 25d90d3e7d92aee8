"""Times the pieces of one V-learner step (HIP events) with the fused and the per-layer forward."""
import sys, ctypes as C
sys.path.insert(0, ".")
import torch
import bench
from pql_amd import _lib as L
from pql_amd.models.mlp import mlp_forward_raw

def timeit(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for fused in (True, False):
    sys.argv = ["bench.py", "--no-streams"] + ([] if fused else ["--no-fused"])
    args = bench.parse()
    dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
    cfg, env, actor, v, p = bench.build_system(args, 0, 1, dev, None)
    bench.prefill(actor, v, p, env, cfg, args, dev)
    ws = v._workspace(8192); B = 8192; al, cl = v.actor.layout, v.critic.layout; O = v.memory.ring.O
    idx = torch.randint(v.memory.cur_capacity, (B,), device=dev)
    v._step_kernels(ws, idx, ws["draw"].normal_())
    t = {}
    t["actor"] = timeit(lambda: mlp_forward_raw(al, v.actor.arena.data, ws["xn_obs"], L.ACT_TANH_NOISE, ws["draw"], 0.8, 0.2, ws["acts_a"], ws["xn_sa"][:, O:], packed=v.pk_actor, stash_all=False))
    t["target"] = timeit(lambda: mlp_forward_raw(cl, v.critic_target.arena.data, ws["xn_sa"], L.ACT_NONE, acts=ws["acts_t"], packed=v.pk_target, stash_all=False))
    t["critic"] = timeit(lambda: mlp_forward_raw(cl, v.critic.arena.data, ws["x_sa"], L.ACT_NONE, acts=ws["acts_c"], packed=v.pk_critic, stash_all=True))
    t["bwd"] = timeit(lambda: L.check(L.lib.pqlk_mlp_backward(C.byref(cl.desc), L.ptr(v.critic.arena.data), L.ptr(ws["x_sa"]), ws["ld_sa"], B,
                                        L.ptr(ws["acts_c"]), L.ptr(ws["dy"]), L.ptr(ws["grads"]), ws["splits"], None, 0, 0, 0,
                                        None, 0, L.ptr(ws["bwd"]), ws["bwd"].numel(), L.stream(dev))))
    if fused:
        t["pack2"] = timeit(lambda: (v.pk_critic.refresh(v.critic.arena.data), v.pk_target.refresh(v.critic_target.arena.data)))
    t["whole_step"] = timeit(lambda: v._step_kernels(ws, idx, ws["draw"]))
    print("fused" if fused else "per-layer", {k: round(x, 1) for k, x in t.items()})
