#!/bin/bash
# median duration per k_gemm instantiation and grid in a V-only (or P-only) eager trace: tools/debug/gemm_times.sh --v-only
bash tools/debug/trace_step.sh "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, statistics, collections
f = glob.glob('gpurun_out/trace_step/t/*kernel_trace.csv')[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "k_gemm" in n or "k_mlp_fwd_fused" in n:
        d[(n[:44], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    print("%-44s grid %5s x %4s x %3s  n=%3d  median %7.2f us  min %7.2f" % (*k, len(v), statistics.median(v), min(v)))
PY
