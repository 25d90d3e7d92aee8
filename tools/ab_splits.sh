for sp in 8 12 16 24 32; do
  for mode in --v-only --p-only; do
    PQLK_SPLITS=$sp python bench.py --no-cpu-baseline --repeat 3 --steps 300 --warmup 30 $mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('splits $sp [$mode]', round(d['value'],1), 'median', round(d['repeats']['median'],1), 'gemm_ms', round(d['roofline']['ms_per_launch_group'],4))"
  done
done
