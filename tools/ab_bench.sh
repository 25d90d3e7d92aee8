#!/bin/bash
# A/B on ONE box (box-to-box spread on this pool is ~10 %): usage tools/ab_bench.sh <tag> "<flagsA>" "<flagsB>" ...
TAG=$1; shift
mkdir -p gpurun_out/$TAG
for flags in "$@"; do
  for mode in --v-only --p-only ""; do
    python bench.py --no-cpu-baseline --repeat 3 --steps 300 --warmup 30 $mode $flags 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$mode] [$flags]', round(d['value'],1), 'median', round(d['repeats']['median'],1), 'gemm_ms', round(d['roofline']['ms_per_launch_group'],4), 'gather_us', round(d['roofline_gather']['us_per_launch'],2))"
  done
done
