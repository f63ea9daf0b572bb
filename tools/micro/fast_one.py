"""One KITTI image through vo_fast_detect ten times (k_fast_nms on 240 tiles of level 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from PIL import Image
from visual_odometry_amd import api
im = np.ascontiguousarray(np.array(Image.open("tests/golden/kitti/000000.png")))
ctx = api.VoContext(0)
for _ in range(10):
    kp = ctx.fast_detect(im, 20, True)
print(len(kp))
ctx.close()
