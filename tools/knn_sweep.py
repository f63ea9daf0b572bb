import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from visual_odometry_amd import api, synth
ctx = api.VoContext(0, max_features=16384) if True else None
ctx.prof_enable(True)
for n in (1000, 2000, 3000, 4096, 6000, 8192):
    q, t = synth.matcher_descriptors(n)
    row = []
    for s in (0, 1, 2, 3, 4, 6, 8, 12, 16):
        ctx.set_option(8, s)
        for _ in range(3):
            ctx.knn2_hamming(q, t)
        ctx.prof_get(0, True)
        for _ in range(20):
            ctx.knn2_hamming(q, t)
        k, ms = ctx.prof_get(0)
        row.append("%d:%.1f" % (s, ms / k * 1e3))
    print(n, " ".join(row))
