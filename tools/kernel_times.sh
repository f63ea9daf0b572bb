#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of one-stream bench.py batches, summarised per kernel.
# usage: tools/kernel_times.sh <tag> [extra bench.py args]
set -e
TAG=${1:-kt}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp VO_SYNTH_CACHE=/tmp/vo_synth_cache VO_SYNTH_NOFORK=0
ARGS="--streams 1 --frames 49 --base-frames 49 --no-cpu-baseline --no-8k --no-pcie --no-default-queues --no-configs4 --no-single-pair --no-lk $*"
cd $GRAFT_REPO_ROOT
timeout -k 10 240 python3 bench.py $ARGS --steps 1 --warmup 0 > $OUT/render.log 2>&1
export VO_SYNTH_NOFORK=1
cd /tmp
timeout -k 10 240 rocprofv3 --kernel-trace -d $OUT/trace -o trace -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS --steps 10 --warmup 2 > $OUT/trace.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py $OUT/trace/trace_results.db 0.3 > $OUT/kernels.txt
cat $OUT/kernels.txt
