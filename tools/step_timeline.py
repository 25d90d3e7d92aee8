#!/usr/bin/env python3
"""Timeline of ONE learner step from a rocprofv3 kernel trace: every launch of a steady-state step with its duration and the
gap to its predecessor (both on the device clock).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o tl -- python3 bench.py --p-only --no-streams --steps 40 \
        --warmup 8 --repeat 1 --burn-in-ms 0 --no-roofline --no-cpu-baseline
    python tools/step_timeline.py gpurun_out/tl k_adamw

The step is delimited by the kernel named last (one launch per step: the optimiser); the median over the steady-state steps is
printed per position in the step."""
import csv
import glob
import statistics
import sys


def main():
    import os
    arg = sys.argv[1]
    path = sorted(glob.glob(os.path.join(arg, "**", "*kernel_trace.csv"), recursive=True) if os.path.isdir(arg) else glob.glob(arg))[0]
    last = sys.argv[2] if len(sys.argv) > 2 else "k_adamw"
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    steps, cur = [], []
    for r in rows:
        cur.append(r)
        if r["Kernel_Name"].startswith(last):
            steps.append(cur)
            cur = []
    # steady state: the most common launch count, steps from the second half of the trace
    n = statistics.mode(len(s) for s in steps[len(steps) // 2:])
    good = [s for s in steps[len(steps) // 2:] if len(s) == n]
    print(f"{len(good)} steps of {n} launches ({path})")
    tot_k = tot_g = 0.0
    for i in range(n):
        name = good[0][i]["Kernel_Name"].split("(")[0][:70]
        dur = statistics.median((int(s[i]["End_Timestamp"]) - int(s[i]["Start_Timestamp"])) / 1e3 for s in good)
        gap = statistics.median((int(s[i]["Start_Timestamp"]) - int(s[i - 1]["End_Timestamp"])) / 1e3 for s in good) if i else 0.0
        grid = good[0][i].get("Grid_Size_X", "?")
        wg = good[0][i].get("Workgroup_Size_X", "?")
        tot_k += dur
        tot_g += gap
        print(f"{i:3d} {name:70s} grid {grid:>8s}/{wg:>4s}  {dur:8.1f} us   gap {gap:6.1f}")
    print(f"kernels {tot_k:.1f} us + gaps {tot_g:.1f} us = {tot_k + tot_g:.1f} us per step")


if __name__ == "__main__":
    main()
