"""The bundle-adjustment FRONT END of the host mirror (reference: src/sfm/bundle_adjustment.cpp:29-130,
SURVEY.md 8f row 3): addKeyFrame over a sliding window -> matcher() -> createMatchMatrix() -> setPBAPoints(),
against the oracle stage by stage and the reference's host algorithms restated here."""
import numpy as np
import pytest

from tests.test_host_tracks import reference_match_matrix
from visual_odometry_amd import host_api, synth

pytestmark = pytest.mark.gpu

NF = 1200


@pytest.mark.parametrize("descriptor", ["akaze", "orb"])  # akaze: what the reference's addKeyFrame calls (:41)
def test_key_frame_window_tracks_and_points(oracle, descriptor):
    fr, poses, K = synth.make_sequence(5, 6)
    ba = host_api.HostBundleAdjustment(K, max_frames=4, nfeatures=NF, descriptor=descriptor)
    for k in range(6):
        ba.add_key_frame(fr[k], poses[k][:, :3], poses[k][:, 3])
    nf, npairs, nrows, npts = ba.counts()
    assert nf == 4  # the window slid twice: key frames 2..5
    win = list(range(2, 6))
    if descriptor == "orb":
        po = oracle.orb_params(NF)
        feats = [oracle.orb_detect_compute(fr[k], po) for k in win]
    else:
        feats = [oracle.akaze_detect_compute(fr[k], oracle.akaze_params()) for k in win]
    for c in range(4):
        kp, desc = ba.features(c)
        assert np.array_equal(kp, feats[c][0]) and np.array_equal(desc, feats[c][1])
    # matcher(): pairs (i, i+1), (i, i+2); knn2 + ratio + gate + findEssentialMat(RANSAC)
    pairs = []
    p = 0
    for i in range(4):
        for j in range(i + 1, min(4, i + 3)):
            knn, nv = oracle.knn2_hamming(feats[i][1], feats[j][1])
            good = oracle.match_filter(knn, nv, feats[i][0], feats[j][0], 0.7, 200.0)
            if len(good) < 5:
                continue
            p0 = np.stack([feats[i][0]["x"][good["queryIdx"]], feats[i][0]["y"][good["queryIdx"]]], 1)
            p1 = np.stack([feats[j][0]["x"][good["trainIdx"]], feats[j][0]["y"][good["trainIdx"]]], 1)
            rc, E, mask, _ = oracle.find_essential(p0, p1, K, oracle.RANSAC)
            src, dst, m, inl = ba.pair(p)
            assert (src, dst) == (i, j) and np.array_equal(m, good) and np.array_equal(inl, mask)
            pairs.append((i, j, good["queryIdx"], good["trainIdx"], mask))
            p += 1
    assert p == npairs == 5
    assert all(len(q[2]) > 50 for q in pairs)
    # createMatchMatrix: the reference's loops on the same pairs
    want = reference_match_matrix(pairs, 4)
    mm = ba.match_matrix()
    assert mm.shape == want.shape and np.array_equal(mm, want) and nrows == len(want)
    # setPBAPoints: tracks with >= 3 views, triangulated from their first two views, then the three filters
    P = [oracle.projection_matrix(K, np.hstack([poses[k][:, :3], poses[k][:, 3:4]])) for k in win]
    pts, vis, img = ba.points()
    exp = []
    for row in want:
        cams = [c for c in range(4) if row[c] != -1]
        if len(cams) < 3:
            continue
        xy = [np.array([[feats[c][0]["x"][row[c]], feats[c][0]["y"][row[c]]]], np.float32) for c in cams]
        X = oracle.triangulate(xy[0], xy[1], P[cams[0]], P[cams[1]])[0]
        R0, t0 = poses[win[cams[0]]][:, :3], poses[win[cams[0]]][:, 3]
        o = R0.T @ (X - t0)
        if np.linalg.norm(o) < 200 and o[2] > 10 and abs(o[0]) < 40:
            exp.append((X, cams, xy))
    assert len(exp) == npts and npts > 20
    for i, (X, cams, xy) in enumerate(exp):
        assert np.allclose(pts[i], X, rtol=1e-6, atol=1e-6)
        assert vis[i].tolist() == [1 if c in cams else 0 for c in range(4)]
        for c, q in zip(cams, xy):
            assert np.array_equal(img[i, c], q[0])
    # the points are where the scene is: reprojection into the first view stays within a few pixels
    Xh = np.hstack([pts, np.ones((len(pts), 1))])
    err = []
    for i, (X, cams, xy) in enumerate(exp):
        u = P[cams[0]] @ Xh[i]
        err.append(np.linalg.norm(u[:2] / u[2] - xy[0][0]))
    assert np.median(err) < 2.0
    ba.close()
