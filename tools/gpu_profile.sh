#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats of bench.py plus two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE cannot share a pass on gfx950), all written under gpurun_out/<tag>/.
# tools/summarize_prof.py turns them into the committed summaries under profiles/.
set -e
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH_ARGS="--streams 1 --frames 49 --steps 5 --warmup 1 --no-cpu-baseline --no-8k"
echo "[profile] kernel trace"; date +%T > $OUT/progress.txt
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py $BENCH_ARGS > $OUT/trace.log 2>&1
echo "[profile] FETCH_SIZE pass"; date +%T >> $OUT/progress.txt
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 bench.py --streams 1 --frames 49 --steps 2 --warmup 1 --no-cpu-baseline --no-8k > $OUT/pmc_fetch.log 2>&1
echo "[profile] WRITE_SIZE pass"; date +%T >> $OUT/progress.txt
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 bench.py --streams 1 --frames 49 --steps 2 --warmup 1 --no-cpu-baseline --no-8k > $OUT/pmc_write.log 2>&1
find $OUT -name '*.csv' | head -30
