import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as o
from visual_odometry_amd import api, synth
fr, poses, K = synth.make_sequence(0, 3)
ctx = api.VoContext(0)
kg, dg = ctx.orb_detect_compute(fr[1], api.orb_params(2000))
po = o.orb_params(2000)
for level in range(8):
    for which, bl in ((0, False), (1, True)):
        g = ctx.debug_read_level(which, 0, level); r = o.orb_level(fr[1], po, level, blurred=bl)
        d = np.argwhere(g != r)
        print(level, which, g.shape, len(d), d[:5].tolist(), [(int(g[y,x]), int(r[y,x])) for y,x in d[:5]])
