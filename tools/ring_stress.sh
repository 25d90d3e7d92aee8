#!/bin/bash
# BASELINE configs[4] at HBM scale: "replay 5 M x (Humanoid-shape obs=108 act=21), batch 32768 -- HBM-bound gather / 288 GB buffer
# stress".  5 M rows of 1 KiB are 5 GB; this script takes the same launch and the same job to rings that fill most of the card:
#   1. the HBM-scale round trip (tests/test_fullsize_properties_gpu.py: 64-bit offsets past 2^37 bytes)
#   2. the 8-batch gather launch of cfg #5 over rings of 5 M .. 200 M rows (tools/bench_gather.py NAME@ROWS --auto)
#   3. the whole job (bench.py) with replay 5 M / 50 M / 140 M rows (V ring 1 KiB + P obs ring 512 B per row: 140 M = 215 GB)
#     gpurun --timeout 1100 -- 'bash tools/ring_stress.sh r04_h'      -> gpurun_out/<tag>/ring_stress.log
set -eo pipefail
TAG=${1:-stress}
OUT=gpurun_out/$TAG
mkdir -p $OUT
LOG=$OUT/ring_stress.log
: > $LOG
echo "[stress] round trip at HBM scale"
timeout -k 10 300 python3 -m pytest tests/test_fullsize_properties_gpu.py -q -m gpu -k hbm_scale 2>&1 | tail -3 | tee -a $LOG
for n in cfg5x8 cfg5x8@40M cfg5x8@100M cfg5x8@150M cfg5x8@200M p5x4 p5x4@200M; do
  echo "[stress] gather $n"
  timeout -k 10 300 python3 tools/bench_gather.py $n --auto 2>&1 | tee -a $LOG
done
for rows in 5000000 50000000 140000000; do
  echo "[stress] bench.py replay $rows"
  timeout -k 10 400 python3 bench.py --no-cpu-baseline --task Humanoid --batch 32768 --nstep 5 --replay $rows --steps 200 > $OUT/bench_cfg5_replay$rows.json 2> $OUT/bench_cfg5_replay$rows.err
  python3 - $OUT/bench_cfg5_replay$rows.json <<'P' | tee -a $LOG
import json, sys
d = json.load(open(sys.argv[1]))
g = d.get("roofline_gather") or {}
print(f"bench.py cfg5 replay {d['config']['per_rank']['replay_rows']:>11} rows: value {d['value']:.1f} (median {d['repeats']['median']:.1f}) V steps/s, "
      f"MFMA frac {d['roofline']['frac']:.3f}, K-batch gather {g.get('us_per_launch', float('nan')):.1f} us = {g.get('frac', float('nan')):.3f} of 8 TB/s")
P
done
echo "[stress] done"
