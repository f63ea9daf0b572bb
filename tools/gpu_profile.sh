#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats of bench.py plus separate PMC passes
# (FETCH_SIZE, WRITE_SIZE and the SQ counters cannot share a pass on gfx950), all written under
# gpurun_out/<tag>/.  tools/summarize_prof.py turns them into the committed summaries under profiles/.
# The frames are rendered once, unprofiled, into VO_SYNTH_CACHE: rocprofv3's preloaded library initialises
# the GPU before main, after which the renderer must not fork.
set -e
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp VO_SYNTH_CACHE=/tmp/vo_synth_cache VO_SYNTH_NOFORK=0
ARGS="--streams 1 --frames 49 --base-frames 49 --no-cpu-baseline --no-8k --no-pcie --no-default-queues --no-configs4 --no-single-pair --no-lk"
echo "[profile] render"; date +%T > $OUT/progress.txt
timeout -k 10 240 python3 bench.py $ARGS --steps 1 --warmup 0 > $OUT/render.log 2>&1
export VO_SYNTH_NOFORK=1
echo "[profile] kernel trace"; date +%T >> $OUT/progress.txt
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py $ARGS --steps 5 --warmup 1 > $OUT/trace.log 2>&1
echo "[profile] FETCH_SIZE pass"; date +%T >> $OUT/progress.txt
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 bench.py $ARGS --steps 2 --warmup 1 > $OUT/pmc_fetch.log 2>&1
echo "[profile] WRITE_SIZE pass"; date +%T >> $OUT/progress.txt
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 bench.py $ARGS --steps 2 --warmup 1 > $OUT/pmc_write.log 2>&1
echo "[profile] SQ pass"; date +%T >> $OUT/progress.txt
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -o sq -- python3 bench.py $ARGS --steps 2 --warmup 1 > $OUT/pmc_sq.log 2>&1
find $OUT -name '*.csv' | head -30
