export VO_SYNTH_CACHE=/tmp/vc
OUT=${1:-gpurun_out/sweep}
mkdir -p $OUT
for cfg in ${SWEEP:-"3 48" "2 48" "4 48" "6 48" "3 32" "4 32" "6 32" "3 64" "4 64" "3 96" "2 96"}; do
  set -- $cfg
  timeout -k 10 120 python bench.py --streams $1 --chunk $2 --steps 6 --warmup 1 --no-cpu-baseline --no-8k --no-pcie --no-default-queues > $OUT/s$1_c$2.log 2>/dev/null
  python -c "
import json,sys
j=json.loads(open('$OUT/s$1_c$2.log').read().strip().splitlines()[-1])
print('streams $1 chunk $2: %.0f pairs/s  host queueing %.1f ms waiting %.1f ms'%(j['value'], j['host_ms_last_sequence']['queueing'], j['host_ms_last_sequence']['waiting']))
"
done
