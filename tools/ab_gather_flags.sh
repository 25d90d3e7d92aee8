set -e
mkdir -p gpurun_out/r4b
export TMPDIR=/tmp
python -m pytest tests/test_handoff_gpu.py -m gpu -x -q > gpurun_out/r4b/tests.log 2>&1 || { tail -40 gpurun_out/r4b/tests.log; exit 1; }
tail -2 gpurun_out/r4b/tests.log
for F in 3 7 11 15; do
  PQL_GATHER_FLAGS=$F python bench.py --no-cpu-baseline --repeat 1 > gpurun_out/r4b/bench_f$F.json 2> gpurun_out/r4b/bench_f$F.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r4b/bench_f$F.json"))
g=d["roofline_gather"]; p=d.get("roofline_gather_p",{})
print("flags $F value %.1f  V gather in-step %.2f us (%.3f)  read-sweep %.2f  b2b %.2f | P gather %.2f us (%.3f) b2b %.2f" % (d["value"], g["us_per_launch"], g["frac"], g["us_per_launch_behind_read_sweep"], g["us_per_launch_back_to_back"], p.get("us_per_launch",0), p.get("frac",0), p.get("us_per_launch_back_to_back",0)))
PY
done
for F in 3 7; do
  PQL_GATHER_FLAGS=$F rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4b/v_f$F -o v -- python3 bench.py --no-cpu-baseline --repeat 1 --burn-in-ms 0 --no-roofline --steps 200 --warmup 24 --no-streams --v-only > /dev/null 2> gpurun_out/r4b/v_f$F.err
  f=$(find gpurun_out/r4b/v_f$F -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r4b/v_f${F}_kernel_stats.csv; rm -rf gpurun_out/r4b/v_f$F
  grep -i "gather\|philox" gpurun_out/r4b/v_f${F}_kernel_stats.csv | cut -c1-200
done
PQL_GATHER_FLAGS=3 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4b/p_f3 -o p -- python3 bench.py --no-cpu-baseline --repeat 1 --burn-in-ms 0 --no-roofline --steps 200 --warmup 24 --no-streams --p-only > /dev/null 2> gpurun_out/r4b/p_f3.err
f=$(find gpurun_out/r4b/p_f3 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r4b/p_f3_kernel_stats.csv; rm -rf gpurun_out/r4b/p_f3
grep -i "gather\|philox" gpurun_out/r4b/p_f3_kernel_stats.csv | cut -c1-200
