#!/usr/bin/env python3
"""Probe: does the NEXT step's actor forward (needs neither the critic's new weights nor the Polyak target) hide the V step's
low-utilisation tail (layer-1 dW, slab sum, AdamW) when forked onto a second stream inside one 8-step hipGraph?

    python tools/probes/overlap_probe.py [bench.py flags]

Prints us per V step for the serial order and for the pipelined order (fork after the layer-2 dX), same launches in both."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from pql_amd import _lib as L  # noqa: E402
from pql_amd.models.mlp import mlp_forward_raw  # noqa: E402


def main():
    sys.argv = ["bench.py", "--no-cpu-baseline"] + sys.argv[1:]
    args = bench.parse()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    torch.manual_seed(42)
    cfg, env, actor, v, p = bench.build_system(args, 0, 1, dev, None)
    bench.prefill(actor, v, p, env, cfg, args, dev)
    B = int(cfg.algo.batch_size)
    v.learn()
    torch.cuda.synchronize()
    ws = dict(v._ws, **v._ws["slots"][0])
    al, cl = v.actor.layout, v.critic.layout
    O = v.memory.ring.O
    nl = cl.n_layers
    v._fused_tail = False   # the ranged backward leaves no norm partials: AdamW runs its own norm pass (both orders alike)
    gamma_n = float(cfg.algo.gamma) ** int(cfg.algo.nstep)
    K = 8
    fork_at = int(os.environ.get("FORK_AT", "1"))   # the tail that runs beside the actor forward starts with dW of layers < fork_at

    def actor_fwd():
        mlp_forward_raw(al, v.actor.arena.data, ws["xn_sa"], L.ACT_TANH_NOISE, v._ahead.normal[0], 0.8, 0.2, ws["acts_a"], ws["xn_sa"][:, O:],
                        packed=v.pk_actor, stash_all=False)

    def fwd():
        mlp_forward_raw(cl, v.critic_target.arena.data, ws["xn_sa"], L.ACT_NONE, acts=ws["acts_t"], packed=v.pk_target, stash_all=False)
        mlp_forward_raw(cl, v.critic.arena.data, ws["x_sa"], L.ACT_NONE, acts=ws["acts_c"], packed=v.pk_critic, stash_all=True)

    def bwd(hi, lo):
        L.check(L.lib.pqlk_mlp_backward_layers(C.byref(cl.desc), L.ptr(v.critic.arena.data), L.ptr(ws["x_sa"]), ws["ld_sa"], B, L.ptr(ws["acts_c"]),
                                               None, L.ptr(ws["acts_t"]), L.ptr(ws["rew"]), L.ptr(ws["done"]), gamma_n, L.ptr(ws["scratch"]),
                                               L.ptr(ws["grads"]), ws["splits"], L.ptr(ws["bwd"]), ws["bwd"].numel(), hi, lo, L.stream(dev)))

    def serial():
        for _ in range(K):
            actor_fwd(); fwd(); bwd(nl - 1, fork_at); bwd(fork_at - 1, 0); v._step_post(ws)

    side = torch.cuda.Stream(dev)

    def pipelined():
        actor_fwd()
        for k in range(K):
            fwd(); bwd(nl - 1, fork_at)
            cur = torch.cuda.current_stream(dev)
            if k + 1 < K:
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    actor_fwd()
            bwd(fork_at - 1, 0); v._step_post(ws)
            if k + 1 < K:
                cur.wait_stream(side)

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

    graphs = {"serial": capture(serial), "pipelined": capture(pipelined)}
    res = {k: [] for k in graphs}
    for r in range(7):
        for name, g in graphs.items():
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                g.replay()
            e1.record(); e1.synchronize()
            res[name].append(e0.elapsed_time(e1) / (10 * K) * 1e3)
    for name, xs in res.items():
        xs.sort()
        print(f"{name:10s} fork_at={fork_at}: median {xs[len(xs) // 2]:.1f} us / V step   min {xs[0]:.1f}")


if __name__ == "__main__":
    main()
