"""GPU tests of the sequence runner (vo_seq_run, BASELINE.json configs[3]): a sequence cut into
chunks over several streams must give, pair for pair, exactly what one vo_pairs_run_dev call over
the same frames gives (which tests/test_gpu_pipeline.py pins to the oracle), for device, pageable
and strided host input; and the first pairs are checked against the oracle directly."""
import ctypes as C

import numpy as np
import pytest
import torch

from visual_odometry_amd import api, synth

pytestmark = pytest.mark.gpu

K = synth.KITTI_K
W, H = 640, 192
KS = K.copy()
KS[0, 2], KS[1, 2] = 320.0, 96.0


def as_tuple(r):
    return (r.n_kp0, r.n_kp1, r.n_good, r.n_inliers, r.n_pose, r.status, tuple(r.E), tuple(r.R), tuple(r.t))


@pytest.fixture(scope="module")
def seq_frames():
    fr, poses, _ = synth.make_sequence(3, 12, w=W, h=H, K=KS)
    return fr, poses


@pytest.fixture(scope="module")
def batch_results(seq_frames):
    fr, _ = seq_frames
    ctx = api.VoContext(0)
    d = torch.from_numpy(fr).to("cuda:0")
    torch.cuda.synchronize()
    n = fr.shape[0]
    res = ctx.pairs_run_dev(d.data_ptr(), n, W, H, W, W * H, KS, api.pair_params(500))
    out = [as_tuple(r) for r in res]
    ctx.close()
    return out


@pytest.mark.parametrize("streams,chunk", [(1, 11), (2, 4), (3, 3), (4, 1), (2, 32)])
def test_seq_device_equals_batch(seq_frames, batch_results, streams, chunk):
    fr, _ = seq_frames
    d = torch.from_numpy(fr).to("cuda:0")
    torch.cuda.synchronize()
    n = fr.shape[0]
    seq = api.VoSequence(0, n_streams=streams, pairs_per_chunk=chunk)
    for _ in range(2):  # second pass: contexts are reused with work history
        res = seq.run(d.data_ptr(), n, W, H, W, W * H, KS, api.pair_params(500))
        assert [as_tuple(r) for r in res] == batch_results
    seq.close()


def test_seq_host_pageable_and_strided(seq_frames, batch_results):
    fr, _ = seq_frames
    n = fr.shape[0]
    seq = api.VoSequence(0, n_streams=2, pairs_per_chunk=4)
    res = seq.run(np.ascontiguousarray(fr), n, W, H, W, W * H, KS, api.pair_params(500), mem=api.MEM_HOST)
    assert [as_tuple(r) for r in res] == batch_results
    # rows padded to 704 bytes, frames padded by 3 rows
    pad = np.full((n, H + 3, 704), 201, np.uint8)
    pad[:, :H, :W] = fr
    res = seq.run(pad, n, W, H, 704, 704 * (H + 3), KS, api.pair_params(500), mem=api.MEM_HOST)
    assert [as_tuple(r) for r in res] == batch_results
    seq.close()


def test_seq_host_pinned(seq_frames, batch_results):
    fr, _ = seq_frames
    n = fr.shape[0]
    pinned = torch.from_numpy(fr).pin_memory()
    seq = api.VoSequence(0, n_streams=2, pairs_per_chunk=5)
    res = seq.run(pinned.data_ptr(), n, W, H, W, W * H, KS, api.pair_params(500), mem=api.MEM_HOST_PINNED)
    assert [as_tuple(r) for r in res] == batch_results
    seq.close()


def test_seq_against_oracle(seq_frames, oracle):
    fr, poses = seq_frames
    n = fr.shape[0]
    d = torch.from_numpy(fr).to("cuda:0")
    torch.cuda.synchronize()
    seq = api.VoSequence(0, n_streams=2, pairs_per_chunk=2)
    res = seq.run(d.data_ptr(), n, W, H, W, W * H, KS, api.pair_params(500))
    po = oracle.orb_params(500)
    feats = [oracle.orb_detect_compute(f, po) for f in fr[:5]]
    for p in range(4):
        (k0, d0), (k1, d1) = feats[p], feats[p + 1]
        knn, nv = oracle.knn2_hamming(d0, d1)
        good = oracle.match_filter(knn, nv, k0, k1, 0.7, 200.0)
        p0 = np.stack([k0["x"][good["queryIdx"]], k0["y"][good["queryIdx"]]], 1)
        p1 = np.stack([k1["x"][good["trainIdx"]], k1["y"][good["trainIdx"]]], 1)
        rc, E, mask, _ = oracle.find_essential(p0, p1, KS, oracle.RANSAC)
        r = res[p]
        assert (r.n_kp0, r.n_kp1, r.n_good, r.status) == (len(k0), len(k1), len(good), rc)
        if rc == 0:
            Eg = np.array(r.E).reshape(3, 3)
            assert min(np.abs(Eg - E).max(), np.abs(Eg + E).max()) < 1e-6
            assert r.n_inliers == int(mask.sum())
            ng, R, t, pm, _ = oracle.recover_pose(E, p0, p1, KS, 200.0, mask)
            assert r.n_pose == ng
            assert np.abs(np.array(r.R).reshape(3, 3) - R).max() < 1e-6  # north_star: 1e-6
            assert np.abs(np.array(r.t) - t).max() < 1e-6
    seq.close()


def test_seq_bad_arguments(seq_frames):
    fr, _ = seq_frames
    seq = api.VoSequence(0, n_streams=2, pairs_per_chunk=4)
    with pytest.raises(api.VoError):
        seq.run(np.ascontiguousarray(fr), 1, W, H, W, W * H, KS, api.pair_params(500), mem=api.MEM_HOST)
    with pytest.raises(api.VoError):
        seq.run(np.ascontiguousarray(fr), 4, W, H, W - 1, W * H, KS, api.pair_params(500), mem=api.MEM_HOST)
    with pytest.raises(api.VoError):
        seq.run(np.ascontiguousarray(fr), 4, W, H, W, W * H, KS, api.pair_params(500), mem=7)
    seq.close()
