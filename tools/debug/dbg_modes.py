import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import test_handoff_gpu as t
distl = len(sys.argv) > 1 and sys.argv[1] == "distl"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
runs = {
 "serial_a": dict(no_graph=True, no_streams=True),
 "serial_b": dict(no_graph=True, no_streams=True),
 "graph_only": dict(no_streams=True),
 "streams_only": dict(no_graph=True),
 "fast": dict(),
}
res = {k: t._run_schedule(steps, distl=distl, **kw) for k, kw in runs.items()}
base = res["serial_a"]
for name, r in res.items():
    bad = []
    for k, a in base.items():
        if k == "counts": continue
        if not torch.equal(a, r[k]):
            d = (a.double() - r[k].double()).abs()
            bad.append((k, float(d.max()), int((d > 0).sum())))
    print(name, r["counts"], bad)
