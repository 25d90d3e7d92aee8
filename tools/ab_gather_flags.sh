#!/bin/bash
# usage: tools/ab_gather_flags.sh <tag>      (run on the GPU box from the repo root)
# A/B of the replay gather's flag word (include/pqlk.h PQLK_GATHER_*: 3 = default, +4 non-temporal record loads, +8 non-temporal
# tile stores) in the learner's own launches: the bench line's three timings of the gather per flag word, then the per-kernel
# average of rocprofv3's trace of the V-only run.  Round 4: profiles/README.md (r04), DESIGN.md section 11.
TAG=${1:-ab_gather}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for F in 3 7 11 15; do
  PQL_GATHER_FLAGS=$F python bench.py --no-cpu-baseline --repeat 1 > $OUT/bench_f$F.json 2> $OUT/bench_f$F.err || { tail -5 $OUT/bench_f$F.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$OUT/bench_f$F.json"))
g=d["roofline_gather"]; p=d.get("roofline_gather_p",{})
print("flags $F value %.1f  V gather own launch %.2f us (%.3f)  added to the queue %.2f  back to back %.2f | P gather %.2f us (%.3f)" % (d["value"], g["us_per_launch"], g["frac"], g["us_added_to_the_v_queue"], g["us_per_launch_back_to_back"], p.get("us_per_launch",0), p.get("frac",0)))
PY
done
for F in 3 7 15; do
  PQL_GATHER_FLAGS=$F rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/v_f$F -o v -- python3 bench.py --no-cpu-baseline --repeat 1 --burn-in-ms 0 --no-roofline --steps 200 --warmup 24 --no-streams --v-only > /dev/null 2> $OUT/v_f$F.err
  f=$(find $OUT/v_f$F -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/v_f${F}_kernel_stats.csv; rm -rf $OUT/v_f$F
  echo "flags $F, kernel trace: $(grep -h gather $OUT/v_f${F}_kernel_stats.csv | awk -F'\",' '{print $2}')"
done
