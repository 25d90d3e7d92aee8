"""Per-block step times of the bench schedule right after set-up: python tools/debug/first_steps.py [block] [blocks]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

def main():
    blk = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    extra = sys.argv[3:]
    sys.argv = [sys.argv[0], "--no-cpu-baseline"] + [a for a in extra if a.startswith("--")]
    mode = "v_only" if "v" in extra else "p_only" if "p" in extra else "schedule"
    args = bench.parse()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    torch.manual_seed(42)
    cfg, env, actor, v, p = bench.build_system(args, 0, 1, dev, None)
    critic, policy = bench.prefill(actor, v, p, env, cfg, args, dev)
    sched = bench.Schedule(actor, v, p, env, cfg, dev, critic, policy, mode=mode)
    out = []
    for b in range(nblk):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(blk):
            sched.step()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / blk * 1e3)
    print(" ".join(f"{x:.3f}" for x in out))

main()
