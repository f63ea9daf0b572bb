"""Tracker timing: vo_lk_track on FAST corners of KITTI-shaped frames, and the per-frame driver."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visual_odometry_amd import api, host_api, synth

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 40
frames, poses, K = synth.make_sequence(0, nf, workers=8)
ctx = api.VoContext(0)
kp = ctx.fast_detect(frames[0], 20, True)
p0 = np.stack([kp["x"], kp["y"]], 1).astype(np.float32)
prm = api.lk_params()
for rep in range(3):
    t0 = time.perf_counter()
    p1, st = ctx.lk_track(frames[0], frames[1], p0, prm)
    t1 = time.perf_counter()
    p2, st2 = ctx.lk_track(None, frames[2], p1, prm)
    t2 = time.perf_counter()
    print("lk_track %d points: %.2f ms (both pyramids), %.2f ms (cached prev); tracked %.0f%%" % (
        len(p0), (t1 - t0) * 1e3, (t2 - t1) * 1e3, 100 * st.mean()))
ctx.close()
# first use of a process pays for contexts, the RANSAC sample table and buffer growth: shown once, then the
# steady state over the same frames
for label in ("cold (first frames of the process)", "steady state"):
    vo = host_api.HostVisualOdometry(K, 750, 1, 3)
    t0 = time.perf_counter()
    for f in frames:
        vo.add_image(f)
    dt = time.perf_counter() - t0
    print("driver (FAST -> LK -> LMedS E -> recoverPose), %s: %d frames in %.3f s = %.1f frames/s" % (
        label, nf, dt, nf / dt))
st = vo.state()
print("last pair: %d points, %d pose inliers" % (len(st["points"]), int(st["mask"].sum())))
