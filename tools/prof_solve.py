"""Small driver for profiling the geometry kernels in isolation: a few findEssentialMat calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visual_odometry_amd import api, synth

ctx = api.VoContext(0)
p0, p1, R, t, inl = synth.correspondences(1000, seed=1, inlier_ratio=0.6)
for _ in range(3):
    rc, E, m = ctx.find_essential(p0, p1, synth.KITTI_K, 8)
print(rc, int(m.sum()))
ctx.close()
