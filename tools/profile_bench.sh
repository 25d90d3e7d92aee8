#!/bin/bash
# usage: tools/profile_bench.sh <tag> [full]      (run on the GPU box from the repo root)
# rocprofv3 kernel-trace summaries of bench.py:
#   sched   the headline 1:4:8 schedule, three HIP streams (per-kernel averages include cross-stream contention)
#   v_only  the V-learner alone on ONE stream (--no-streams): every kernel's average is contention-free
#   p_only  the P-learner alone on ONE stream
# and, with `full`, the PMC passes (FETCH_SIZE / WRITE_SIZE / matrix-pipe busy cycles: kernel-trace only, separate runs) reduced by
# tools/pmc_traffic.py and tools/pmc_mfma.py, plus un-profiled bench lines for the other BASELINE shapes.
# Outputs land in gpurun_out/<tag>/; copy what should be judged into profiles/.
set -e
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() {   # name, bench flags...
  local name=$1; shift
  # (--burn-in-ms 0 --no-roofline: nothing but set-up, warm-up and the timed steps launches kernels, so the CSV's call counts are
  #  (steps + warm-up) x launches per step plus the set-up's -- tests/test_profiles_cpu.py checks that)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o $name -- python3 bench.py --no-cpu-baseline --repeat 1 --burn-in-ms 0 --no-roofline "$@" \
      > $OUT/${name}_under_rocprof.json 2> $OUT/${name}.err || { tail -5 $OUT/${name}.err; return 1; }
  f=$(find $OUT/$name -name "*kernel_stats.csv" | head -1)
  cp "$f" $OUT/${name}_kernel_stats.csv
  rm -rf $OUT/$name
}
pmc() {   # counter name
  local c=$1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python3 bench.py --steps 48 --warmup 16 \
      --no-cpu-baseline --no-streams --v-only --repeat 1 --burn-in-ms 0 --no-roofline > /dev/null 2> $OUT/pmc_$c.err || { tail -5 $OUT/pmc_$c.err; return 1; }
  find $OUT/pmc_$c -name "*counter_collection.csv" | head -1
}
run sched --steps 400 --warmup 48
run v_only --steps 200 --warmup 24 --no-streams --v-only
run p_only --steps 200 --warmup 24 --no-streams --p-only
if [ "$2" = "full" ]; then
  F=$(pmc FETCH_SIZE); W=$(pmc WRITE_SIZE)
  python3 tools/pmc_traffic.py "$F" "$W" $OUT/pmc_traffic.json > /dev/null
  rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
  # matrix-pipe utilisation per kernel (a third counter pass, kernel-trace only)
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o pmc -- python3 bench.py \
      --steps 48 --warmup 16 --no-cpu-baseline --no-streams --v-only --repeat 1 --burn-in-ms 0 --no-roofline > /dev/null 2> $OUT/pmc_mfma.err || tail -5 $OUT/pmc_mfma.err
  python3 tools/pmc_mfma.py "$(find $OUT/pmc_mfma -name "*counter_collection.csv" | head -1)" $OUT/pmc_mfma.json > /dev/null
  rm -rf $OUT/pmc_mfma
  echo "[profile] bench lines"
  python3 bench.py --steps 800 --warmup 96 > $OUT/bench.json 2> $OUT/bench.err
  python3 bench.py --no-cpu-baseline --hidden 512,256,128 > $OUT/bench_hidden_ref.json 2>/dev/null
  python3 bench.py --no-cpu-baseline --task ShadowHand --num-envs 16384 --distl --replay 2000000 --hidden 512,256,128 > $OUT/bench_cfg4.json 2>/dev/null
  python3 bench.py --no-cpu-baseline --task ShadowHand --num-envs 16384 --distl --replay 2000000 > $OUT/bench_cfg4_hidden512x512x256.json 2>/dev/null
  python3 bench.py --no-cpu-baseline --task Humanoid --batch 32768 --nstep 5 --replay 5000000 --hidden 512,256,128 --steps 200 > $OUT/bench_cfg5.json 2>/dev/null
  python3 bench.py --no-cpu-baseline --task Humanoid --batch 32768 --nstep 5 --replay 5000000 --steps 200 > $OUT/bench_cfg5_hidden512x512x256.json 2>/dev/null
  python3 bench.py --no-cpu-baseline --gpus 2 --layout split2 --share-gpu > $OUT/bench_split2_one_card.json 2>/dev/null
  echo "[profile] multi-rank rehearsals"
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --gpus 2 --backend gloo --share-gpu --steps 100 2>/dev/null | grep "^{" > $OUT/bench_dp2_gloo_one_card.json
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --gpus 2 --backend gloo --share-gpu --steps 100 --scaling strong 2>/dev/null | grep "^{" > $OUT/bench_dp2_strong_gloo_one_card.json
  PQL_FORCE_DP=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline 2>/dev/null | grep "^{" > $OUT/bench_dp1_rccl_one_rank.json   # (RCCL prints a version banner on stdout)
  echo "[profile] gather sweeps"
  python3 tools/bench_gather.py cfg2x8 p2x4 cfg5x8 cfg4x8 --quick > $OUT/gather_sweep.log 2>&1
  tools/probes/bin/gather_limit_probe > $OUT/gather_limit_probe.log 2>&1 || true
  # per-launch timelines of one steady-state V / P step (device clock)
  for m in v p; do
    rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_$m -o tl -- python3 bench.py --$m-only --no-streams --steps 40 --warmup 8 --repeat 1 \
        --burn-in-ms 0 --no-roofline --no-cpu-baseline > /dev/null 2> $OUT/tl_$m.err
    python3 tools/step_timeline.py $OUT/tl_$m k_adamw > $OUT/${m}_step_timeline.txt
    rm -rf $OUT/tl_$m
  done
  python3 bench.py --no-cpu-baseline --rng torch > $OUT/bench_rng_torch.json 2>/dev/null
  python3 tools/roofline_from_stats.py $OUT/v_only_kernel_stats.csv > $OUT/v_only_roofline.md
  python3 tools/roofline_from_stats.py $OUT/p_only_kernel_stats.csv --p-only > $OUT/p_only_roofline.md
fi
