import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import test_handoff_gpu as t

def run(steps, variant, **kw):
    from pql_amd.utils import handoff
    bench = t._bench()
    dev = torch.device("cuda:0")
    torch.manual_seed(1234)
    args = t._args(distl=True, **kw)
    cfg, env, actor, v, p = bench.build_system(args, 0, 1, dev, None)
    critic, policy = bench.prefill(actor, v, p, env, cfg, args, dev)
    sched = bench.Schedule(actor, v, p, env, cfg, dev, critic, policy)
    trace = []
    for i in range(steps):
        if variant == "no_p":
            orig = p.learn; p.learn = lambda *a, **k: None
            sched.step(); p.learn = orig
        else:
            sched.step()
        if variant == "sync_each":
            torch.cuda.synchronize()
        if variant == "trace":
            torch.cuda.synchronize()
            trace.append((v.critic.arena.data.double().sum().item(), v.loss_ring.clone()))
    out = t._state(actor, v, p)
    out["trace"] = trace
    return out

def diff(a, b):
    bad = []
    for k, x in a.items():
        if k in ("counts", "trace"): continue
        if not torch.equal(x, b[k]):
            d = (x.double() - b[k].double()).abs()
            bad.append((k, float(d.max()), int((d > 0).sum())))
    return bad

serial = run(16, "trace", no_graph=True, no_streams=True)
for name, variant, kw in [("fast", "plain", {}), ("fast_again", "plain", {}), ("fast_sync_each", "sync_each", {}),
                          ("fast_trace", "trace", {})]:
    r = run(16, variant, **kw)
    print(name, diff(serial, r))
    if variant == "trace":
        for i, ((s0, l0), (s1, l1)) in enumerate(zip(serial["trace"], r["trace"])):
            print(i, s0 == s1, torch.equal(l0, l1))
s_nop = run(16, "no_p", no_graph=True, no_streams=True)
f_nop = run(16, "no_p")
print("no_p", diff(s_nop, f_nop))
for n in (8, 9, 10, 12):
    print("steps", n, diff(run(n, "plain", no_graph=True, no_streams=True), run(n, "plain")))
