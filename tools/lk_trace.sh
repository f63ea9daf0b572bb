#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of the single-frame tracking path's steps (tools/time_track_steps.py).
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-lktrace}
mkdir -p $OUT
export TMPDIR=/tmp VO_SYNTH_CACHE=/tmp/vo_synth_cache VO_SYNTH_NOFORK=0
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python3 tools/time_track_steps.py > $OUT/steps.txt 2>&1
export VO_SYNTH_NOFORK=1
cd /tmp
timeout -k 10 240 rocprofv3 --kernel-trace -d $OUT/trace -o trace -- python3 $GRAFT_REPO_ROOT/tools/time_track_steps.py > $OUT/trace.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py $OUT/trace/trace_results.db 0.2 > $OUT/kernels.txt
cat $OUT/steps.txt $OUT/kernels.txt
