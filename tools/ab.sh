#!/bin/bash
# Runs on the GPU box: bench.py with two builds of the library on the SAME box, alternating (boxes differ by 1-3 %).
# The other build is visual_odometry_amd/libvo_hip_old.so (git-ignored; build it from the commit to compare with).
for i in 1 2; do
for v in old new; do
  if [ $v = old ]; then export VO_HIP_LIB=$GRAFT_REPO_ROOT/visual_odometry_amd/libvo_hip_old.so; else unset VO_HIP_LIB; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs4 --no-lk --no-8k --no-single-pair --no-default-queues > gpurun_out/ab_$v.json 2>/dev/null || exit 1
  python - <<PY
import json
j=json.load(open("gpurun_out/ab_$v.json"))
print("$v", round(j["value"]), round(j["pcie_inclusive"]["value"]))
PY
done; done
