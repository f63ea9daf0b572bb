#!/bin/bash
# Builds libvo_hip variants whose k_fast_nms stops after phase 0..3 (fill, compass, arc test, score) into
# tools/micro/variants/ (git-ignored), for tools/micro/fast_phases_run.sh on the GPU box.
set -e
cd "$(dirname "$0")/../../visual_odometry_amd/csrc"
mkdir -p ../../tools/micro/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-result"
for n in 0 1 2 3; do
  /opt/rocm/bin/hipcc $FLAGS -DVO_FAST_STOP=$n -c k_features.hip -o /tmp/k_features_stop$n.o
  objs=$(ls build/*.o | grep -v k_features.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/micro/variants/libvo_hip_stop$n.so $objs /tmp/k_features_stop$n.o
done
ls -la ../../tools/micro/variants
