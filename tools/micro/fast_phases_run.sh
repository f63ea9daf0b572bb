#!/bin/bash
# On the GPU box: SQ_INSTS_VALU of k_fast_nms for the full library and the stop-after-phase variants.
set -e
OUT=gpurun_out/fast_phases
mkdir -p $OUT
export TMPDIR=/tmp
for v in full 0 1 2 3; do
  if [ $v = full ]; then unset VO_HIP_LIB; else export VO_HIP_LIB=$PWD/tools/micro/variants/libvo_hip_stop$v.so; fi
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/$v -o p -- python3 tools/micro/fast_one.py > $OUT/$v.log 2>&1
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$v -o t -- python3 tools/micro/fast_one.py > $OUT/t$v.log 2>&1
  echo "variant $v: $(python3 tools/pmc_kernel.py $OUT/$v/p_counter_collection.csv k_fast_nms) $(grep k_fast_nms $OUT/t$v/t_kernel_stats.csv | cut -d, -f2-4)"
done
