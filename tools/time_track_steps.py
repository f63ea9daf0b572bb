"""Wall time of the single-frame tracking path's steps (median of REPS calls each)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visual_odometry_amd import api, synth

REPS = 60
frames, poses, K = synth.make_sequence(0, 4, workers=4)
ctx = api.VoContext(0)
kp = ctx.fast_detect(frames[0], 20, True)
p0 = np.stack([kp["x"], kp["y"]], 1).astype(np.float32)
prm = api.lk_params()
p1, st = ctx.lk_track(frames[0], frames[1], p0, prm)
ok = st.astype(bool)
a, b = p0[ok], p1[ok]


def med(fn):
    ts = []
    for _ in range(REPS):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6


print("points: %d detected, %d tracked" % (len(p0), len(a)))
print("fast_detect                    %7.1f us" % med(lambda: ctx.fast_detect(frames[0], 20, True)))
print("lk_track (both pyramids)       %7.1f us" % med(lambda: ctx.lk_track(frames[0], frames[1], p0, prm)))
ctx.lk_track(frames[0], frames[1], p0, prm)
print("lk_track (cached prev)         %7.1f us" % med(lambda: ctx.lk_track(None, frames[1], p0, prm)))
print("find_essential LMedS           %7.1f us" % med(lambda: ctx.find_essential(a, b, K, method=api.LMEDS)))
rc, E, m = ctx.find_essential(a, b, K, method=api.LMEDS)
print("recover_pose                   %7.1f us" % med(lambda: ctx.recover_pose(E, a, b, K, 200.0, m)))
print("essential_pose LMedS (fused)   %7.1f us" % med(lambda: ctx.essential_pose(a, b, K, method=api.LMEDS)))
print("find_essential RANSAC          %7.1f us" % med(lambda: ctx.find_essential(a, b, K, method=api.RANSAC)))
ctx.close()
