#!/bin/bash
# Quick loop for detector-kernel work (through gpurun): feature parity tests, then kernel-trace stats of a
# one-stream 49-frame batch under gpurun_out/<tag>/.
set -e
TAG=${1:-qf}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp VO_SYNTH_CACHE=/tmp/vo_synth_cache VO_SYNTH_NOFORK=0
timeout -k 10 400 python3 -u -m pytest tests/test_gpu_features.py tests/test_gpu_real_images.py -x -q -m gpu --timeout 120 --timeout-method thread > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
ARGS="--streams 1 --frames 49 --base-frames 49 --no-cpu-baseline --no-8k --no-pcie --no-default-queues"
timeout -k 10 240 python3 bench.py $ARGS --steps 1 --warmup 0 > $OUT/render.log 2>&1
export VO_SYNTH_NOFORK=1
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 bench.py $ARGS --steps 10 --warmup 1 > $OUT/trace.log 2>&1
python3 - $OUT/trace/t_kernel_stats.csv <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0]
    if float(r["Percentage"]) > 1.0: print("%-28s calls %4s avg %9.1f us  %5s %%" % (n[:28], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
P
