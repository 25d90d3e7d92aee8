#!/bin/bash
# usage: tools/profile_bench.sh <tag>   (run on the GPU box from the repo root)
# rocprofv3 kernel-trace summaries of bench.py: the headline schedule, and the V-learner / P-learner alone on ONE stream
# (--no-streams) so every kernel's average is free of cross-stream contention.  Outputs land in gpurun_out/<tag>/; copy the
# *_kernel_stats.csv files you want judged into profiles/.
set -e
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
run() {   # name, bench flags...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o $name -- python3 bench.py --no-cpu-baseline --repeat 1 "$@" \
      > $OUT/${name}_under_rocprof.json 2> $OUT/${name}.err || { tail -5 $OUT/${name}.err; return 1; }
  f=$(find $OUT/$name -name "*kernel_stats.csv" | head -1)
  cp "$f" $OUT/${name}_kernel_stats.csv
  rm -rf $OUT/$name
}
run sched --steps 400 --warmup 48
run v_only --steps 200 --warmup 24 --no-streams --v-only
run p_only --steps 200 --warmup 24 --no-streams --p-only
