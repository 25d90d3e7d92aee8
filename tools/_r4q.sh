OUT=gpurun_out/r4q; mkdir -p $OUT
python -m pytest tests/test_handoff_gpu.py -m gpu -x -q -k "four_ranks" > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log | cut -c1-300
python tools/bench_gather.py p2x4s --quick > $OUT/gather_p.log 2>&1; grep -v amdgpu $OUT/gather_p.log
