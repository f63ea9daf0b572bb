# SQ counters of k_fast_nms on the 49-frame batch (run through gpurun): LDS side and wait side, one pass each
set -e
export TMPDIR=/tmp VO_SYNTH_CACHE=/tmp/vo_synth_cache
ARGS="--streams 1 --frames 49 --base-frames 49 --no-cpu-baseline --no-8k --no-pcie --no-default-queues"
OUT=gpurun_out/${1:-fast_lds}
mkdir -p $OUT
timeout -k 10 240 python3 bench.py $ARGS --steps 1 --warmup 0 > /dev/null 2>&1
export VO_SYNTH_NOFORK=1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $OUT/p -o p -- python3 bench.py $ARGS --steps 2 --warmup 1 > $OUT/p.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/q -o q -- python3 bench.py $ARGS --steps 2 --warmup 1 > $OUT/q.log 2>&1
for k in ${KERNELS:-k_fast_nms}; do
python3 tools/pmc_kernel.py $OUT/p/p_counter_collection.csv $k
python3 tools/pmc_kernel.py $OUT/q/q_counter_collection.csv $k
done
