#!/bin/bash
# Where do k_dx_slice<16,16>'s ~25 us go?  Builds of the library with -DPQLK_SLICE_SKIP=n (the kernel returns after phase n; see
# narrow.h) under tools/probes/bin/slice_skip<n>/, each run through the P-only kernel trace; prints the kernel's average duration.
#   for n in 1 2 3 4; do d=tools/probes/bin/slice_skip$n; mkdir -p $d; git archive HEAD pql_amd/csrc include | tar -x -C $d;
#     make -C $d/pql_amd/csrc -j8 EXTRA=-DPQLK_SLICE_SKIP=$n; done
#   gpurun -- 'bash tools/probes/slice_phases.sh r04_l'
set -eo pipefail
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/${1:-slice}
mkdir -p $OUT
for v in 1 2 3 4 full; do
  if [ $v = full ]; then unset PQLK_LIB; else export PQLK_LIB=$PWD/tools/probes/bin/slice_skip$v/pql_amd/csrc/libpqlk.so; fi
  rm -rf $OUT/tl
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tl -o tl -- python3 bench.py --p-only --no-streams --steps 40 --warmup 8 \
      --repeat 1 --burn-in-ms 0 --no-roofline --no-cpu-baseline > /dev/null 2> $OUT/tl_$v.err
  f=$(find $OUT/tl -name '*kernel_stats.csv' | head -1)
  echo "== after phase $v" | tee -a $OUT/slice_phases.log
  grep -E 'k_dx_slice|k_dpg_minnet_head' $f | awk -F'","|",|,"' '{print "   ", $1, $2, "calls, avg ns", $4}' | tee -a $OUT/slice_phases.log
done
rm -rf $OUT/tl
