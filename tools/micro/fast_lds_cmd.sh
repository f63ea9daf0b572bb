# LDS-side counters of k_fast_nms on the 49-frame batch (run through gpurun)
set -e
export TMPDIR=/tmp VO_SYNTH_CACHE=/tmp/vo_synth_cache
ARGS="--streams 1 --frames 49 --base-frames 49 --no-cpu-baseline --no-8k --no-pcie"
OUT=gpurun_out/fast_lds
mkdir -p $OUT
timeout -k 10 240 python3 bench.py $ARGS --steps 1 --warmup 0 > /dev/null 2>&1
export VO_SYNTH_NOFORK=1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $OUT/p -o p -- python3 bench.py $ARGS --steps 2 --warmup 1 > $OUT/p.log 2>&1
python3 tools/pmc_kernel.py $OUT/p/p_counter_collection.csv k_fast_nms
python3 tools/pmc_kernel.py $OUT/p/p_counter_collection.csv k_rbrief
python3 tools/pmc_kernel.py $OUT/p/p_counter_collection.csv k_orb_select
