"""Does the ~40-step ramp of the MFMA kernels come back after an idle gap?  python tools/debug/ramp_after_idle.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

def blocks(sched, blk, n):
    out = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(blk):
            sched.step()
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / blk * 1e3)
    return " ".join(f"{x:.3f}" for x in out)

sys.argv = [sys.argv[0], "--no-cpu-baseline"]
args = bench.parse()
dev = torch.device("cuda:0"); torch.cuda.set_device(dev); torch.manual_seed(42)
cfg, env, actor, v, p = bench.build_system(args, 0, 1, dev, None)
critic, policy = bench.prefill(actor, v, p, env, cfg, args, dev)
sched = bench.Schedule(actor, v, p, env, cfg, dev, critic, policy, mode="v_only")
print("after set-up     ", blocks(sched, 4, 14))
for gap in (0.01, 0.1, 1.0):
    time.sleep(gap)
    print(f"after {gap:4.2f} s idle", blocks(sched, 4, 14))
time.sleep(0.1)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.1:
    bench.gemm_section_ms(v, iters=8)
print("idle, then 100 ms of the roofline section", blocks(sched, 4, 10))
time.sleep(0.1)
for _ in range(60):
    sched.step()
print("idle, then 60 untimed steps            ", blocks(sched, 4, 10))
