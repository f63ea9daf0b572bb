"""Matcher kernel timing: 8192 x 8192 single launch and the 32-pair 2k batch (HIP events via vo_prof)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visual_odometry_amd import api, synth

frames, poses, K = synth.make_sequence(0, 49, workers=8)
import torch
d = torch.from_numpy(frames).to("cuda:0")
ctx = api.VoContext(0)
q, t = synth.matcher_descriptors(8192)
ctx.prof_enable(True)
us8 = float("nan")
if not os.environ.get("VO_BENCH_KNN_BATCH_ONLY"):  # counter passes: only the batch launch
    for _ in range(2):
        ctx.knn2_hamming(q, t)
    ctx.prof_get(0, True)
    for _ in range(10):
        ctx.knn2_hamming(q, t)
    n, ms = ctx.prof_get(0)
    us8 = ms / n * 1e3
n_, h, w = frames.shape
prm = api.pair_params(2000)
for _ in range(2):
    ctx.pairs_run_dev(d.data_ptr(), n_, w, h, w, w * h, K, prm)
ctx.prof_get(0, True)
for _ in range(10):
    ctx.pairs_run_dev(d.data_ptr(), n_, w, h, w, w * h, K, prm)
n, ms = ctx.prof_get(0)
us2 = ms / n * 1e3
res = ctx.pairs_run_dev(d.data_ptr(), n_, w, h, w, w * h, K, prm)
nk = [res[0].n_kp0] + [r.n_kp1 for r in res]
dp = sum(nk[i] * nk[i + 1] for i in range(n_ - 1))
print("S=%s  8k: %.1f us (%.2f Tpairs/s)   %dx2k: %.1f us (%.2f Tpairs/s)" % (
    os.environ.get("VO_KNN_SPLITS"), us8, 8192 * 8192 / us8 / 1e6, n_ - 1, us2, dp / us2 / 1e6))
