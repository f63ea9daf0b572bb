timeout -k 10 600 python -u -m pytest tests/test_gpu_features.py tests/test_gpu_pipeline.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -2 && export TMPDIR=/tmp VO_SYNTH_CACHE=/tmp/vo_synth_cache && timeout -k 10 240 python3 bench.py --streams 1 --frames 49 --base-frames 49 --no-cpu-baseline --no-8k --no-pcie --steps 1 --warmup 0 > /dev/null 2>&1 && export VO_SYNTH_NOFORK=1 && timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/l0 -o t -- python3 bench.py --streams 1 --frames 49 --base-frames 49 --no-cpu-baseline --no-8k --no-pcie --steps 5 --warmup 1 > gpurun_out/l0.log 2>&1; python3 -c "
import csv
for r in csv.DictReader(open('gpurun_out/l0/t_kernel_stats.csv')):
    if any(k in r['Name'] for k in ('fast_nms','ic_angle','rbrief','orb_select')):
        print(r['Name'][:24], r['Calls'], float(r['AverageNs'])/1e3)
" && unset VO_SYNTH_NOFORK && timeout -k 10 300 python bench.py --no-cpu-baseline --no-8k --no-pcie 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'])"
