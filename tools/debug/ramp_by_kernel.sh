#!/bin/bash
# duration of each learner kernel as a function of the step index right after set-up (eager V-only trace)
bash tools/debug/trace_step.sh --v-only > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/trace_step/t/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
seq = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    for key in ("k_mlp_fwd_fused<2, 2>", "k_mlp_fwd_fused<1, 2>", "k_gemm<1, 128, 128", "k_gemm<2, 128, 128", "k_adamw", "k_reduce_slabs", "k_replay_gather_fast"):
        if key in n:
            seq[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in seq.items():
    per = 2 if k in ("k_mlp_fwd_fused<2, 2>", "k_gemm<1, 128, 128", "k_gemm<2, 128, 128") else 1
    steps = [sum(v[i:i + per]) / per for i in range(0, len(v) - per + 1, per)]
    pick = [0, 1, 2, 4, 8, 12, 16, 24, 32, 40, 47]
    print("%-24s" % k, " ".join("%6.1f" % steps[i] for i in pick if i < len(steps)), " (n=%d)" % len(steps))
PY
