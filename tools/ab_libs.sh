#!/bin/bash
# A/B of several builds of libpqlk.so on ONE box, V-only / P-only / schedule: tools/ab_libs.sh <a.so> <b.so> ...  ("current" = the in-tree build)
for lib in current "$@"; do
  for mode in --v-only --p-only ""; do
    if [ "$lib" = current ]; then unset PQLK_LIB; else export PQLK_LIB=$(realpath $lib); fi
    python bench.py --no-cpu-baseline --repeat 3 --steps 300 --warmup 30 $mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$mode] [$(basename $lib)]', round(d['value'],1), 'median', round(d['repeats']['median'],1), 'gemm_ms', round(d['roofline']['ms_per_launch_group'],4))"
  done
done
