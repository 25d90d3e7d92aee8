#!/bin/bash
# per-dispatch kernel durations of a few P-only (or V-only) steps: tools/debug/trace_step.sh --p-only
export TMPDIR=/tmp
OUT=gpurun_out/trace_step; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python3 bench.py --no-cpu-baseline --repeat 1 --steps 40 --warmup 8 --no-streams --no-graph "$@" > /dev/null 2> $OUT/err.txt
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last ~60 dispatches before the end of the timed region: print name + duration
names = [(r["Kernel_Name"][:70], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X","?"), r.get("Grid_Size_Y","?"), r.get("Grid_Size_Z","?")) for r in rows]
# find a window with the learner kernels in steady state: take dispatches 60% through the trace
i0 = int(len(names) * 0.6)
for n in names[i0:i0 + 48]:
    print("%-70s %8.2f us  grid %s x %s x %s" % n)
PY
