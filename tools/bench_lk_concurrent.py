"""Default pipeline (FAST -> LK -> LMedS E -> recoverPose) through the C++ driver: S sequences side by side."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("VO_HWQ", "8"))
import numpy as np
from visual_odometry_amd import host_api, synth

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 160
Ss = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8, 16]
frames, poses, K = synth.make_sequence(0, nf, workers=8)
for S in Ss:
    if S == 1:
        host_api.timing_enable(True)
    wall, timed, _ = host_api.run_concurrent(K, [frames] * S, warmup=40)
    line = "S=%2d  %7.0f frames/s  (%.1f us per frame of wall time)" % (S, timed / wall, wall / timed * 1e6)
    if S > 1:
        for G in (1, 2, 4):
            if G > S:
                continue
            wall, timed, _ = host_api.run_concurrent(K, [frames] * S, warmup=40, lockstep=G)
            line += "   lock step x%d: %6.0f f/s" % (G, timed / wall)
    if S == 1:
        line += "  " + " ".join("%s=%.0f" % (k.replace(" ", "_"), v[1]) for k, v in host_api.timing_get().items())
        host_api.timing_enable(False)
    print(line, flush=True)
