"""AKAZE timing: vo_akaze_detect_compute on KITTI-shaped frames against the oracle on one host core."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visual_odometry_amd import api, synth
from oracle import pyoracle as o

frames, _, K = synth.make_sequence(0, 4, workers=4)
ctx = api.VoContext(0)
for rep in range(3):
    ts = []
    for f in frames:
        t0 = time.perf_counter()
        kp, d = ctx.akaze_detect_compute(f)
        ts.append(time.perf_counter() - t0)
    print("vo_akaze_detect_compute 1241x376: %.2f ms per frame (%d keypoints)" % (np.median(ts) * 1e3, len(kp)))
t0 = time.perf_counter()
ko, do = o.akaze_detect_compute(frames[-1], o.akaze_params())
print("oracle (one core, unoptimised): %.1f ms; identical: %s" % ((time.perf_counter() - t0) * 1e3,
                                                                 np.array_equal(kp, ko) and np.array_equal(d, do)))
ctx.close()
