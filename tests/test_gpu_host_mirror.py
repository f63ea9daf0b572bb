"""The C++ host-side mirror of the reference's interfaces (visual_odometry_amd/host/: FeatureDetector,
matcher(), createMatchMatrix(), updatePose(), triangulate(), getProjectionMatrix(), points4DToVec())
driven by tests/cpp/host_mirror_run the way the reference's own test drives them
(src/sfm/test/triangulation_test.cpp:40-66), checked against the CPU oracle: keypoints, descriptors
and matches bit-exact, E / R / t / points within 1e-6 (north_star tolerance)."""
import os
import subprocess

import numpy as np
import pytest

from visual_odometry_amd import api, synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "cpp", "host_mirror_run")


def test_host_mirror_matches_oracle(tmp_path, oracle):
    assert os.path.exists(EXE), "tests/cpp/host_mirror_run is built by __graft_entry__.build()"
    w, h, nfeat = 1241, 376, 2000
    frames, poses, K = synth.make_sequence(3, 2)
    raw = tmp_path / "frames.raw"
    frames.tofile(raw)
    out = tmp_path / "out"
    out.mkdir()
    r = subprocess.run([EXE, str(raw), str(w), str(h), str(nfeat), str(out)], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr

    def load(name, dtype):
        return np.fromfile(out / name, dtype=dtype)

    # FeatureDetector::detectFAST(VOFrame*)
    fast = oracle.fast_detect(frames[0], 20, True)
    pts = load("fast_points.bin", np.float32).reshape(-1, 2)
    assert np.array_equal(pts, np.stack([fast["x"], fast["y"]], 1))
    # FeatureDetector::detectComputeORB
    po = oracle.orb_params(nfeat)
    ko = [oracle.orb_detect_compute(f, po) for f in frames]
    for i in range(2):
        assert np.array_equal(load("kp%d.bin" % i, api.KP_DTYPE), ko[i][0])
        assert np.array_equal(load("desc%d.bin" % i, np.uint8).reshape(-1, 32), ko[i][1])
    # FeatureDetector::detectComputeAKAZE (what BundleAdjustment::addKeyFrame calls)
    ka, da = oracle.akaze_detect_compute(frames[0], oracle.akaze_params())
    assert np.array_equal(load("akaze_kp.bin", api.KP_DTYPE), ka)
    assert np.array_equal(load("akaze_desc.bin", np.uint8).reshape(-1, 61), da)
    # matcher(): knn2 + ratio + gate + findEssentialMat(RANSAC)
    knn, nv = oracle.knn2_hamming(ko[0][1], ko[1][1])
    good = oracle.match_filter(knn, nv, ko[0][0], ko[1][0], 0.7, 200.0)
    assert np.array_equal(load("matches.bin", api.DM_DTYPE), good)
    p0 = np.stack([ko[0][0]["x"][good["queryIdx"]], ko[0][0]["y"][good["queryIdx"]]], 1)
    p1 = np.stack([ko[1][0]["x"][good["trainIdx"]], ko[1][0]["y"][good["trainIdx"]]], 1)
    rc, E_r, mask_r, _ = oracle.find_essential(p0, p1, K, oracle.RANSAC)
    assert rc == 0 and np.array_equal(load("inliers.bin", np.uint8), mask_r)
    # createMatchMatrix (reference: src/matcher/matcher.cpp:55-86): an inlier match joins the first track
    # that already holds either of its ends
    tracks = []
    for m, keep in zip(good, mask_r):
        if keep != 1:
            continue
        q, t = int(m["queryIdx"]), int(m["trainIdx"])
        for row in tracks:
            if row[0] == q or row[1] == t:
                row[0], row[1] = q, t
                break
        else:
            tracks.append([q, t])
    assert load("ntracks.bin", np.int32)[0] == len(tracks)
    # updatePose(): findEssentialMat(LMEDS) + recoverPose + sign/transpose convention + chaining
    rc, E, mask, _ = oracle.find_essential(p0, p1, K, oracle.LMEDS)
    ng, R, t, pm, pts4 = oracle.recover_pose(E, p0, p1, K, 200.0, mask)
    assert load("rc.bin", np.int32)[0] == (0 if ng > 8 else 1)
    Eg = load("E.bin", np.float64).reshape(3, 3)
    assert min(np.abs(Eg - E).max(), np.abs(Eg + E).max()) < 1e-6
    assert np.array_equal(load("mask.bin", np.uint8), pm)
    assert np.abs(load("local_R.bin", np.float64).reshape(3, 3) - R.T).max() < 1e-6
    assert np.abs(load("local_t.bin", np.float64) + t).max() < 1e-6
    p3 = load("points_3d.bin", np.float64).reshape(-1, 3)
    ref3 = (pts4[:3] / pts4[3]).T
    ok = pm.astype(bool)
    assert np.allclose(p3[ok], ref3[ok], rtol=1e-6, atol=1e-6)
    # pose chaining: pose_R = local_R * I, pose_t = scale * local_t with scale in (0, 5] or 1
    pose = load("pose.bin", np.float64).reshape(3, 4)
    assert np.abs(pose[:, :3] - R.T).max() < 1e-6
    s = np.linalg.norm(pose[:, 3])
    assert 0 < s <= 5.0 and np.abs(pose[:, 3] / s + t / np.linalg.norm(t)).max() < 1e-6
    # triangulate(p0, p1, getProjectionMatrix(K, pose0), getProjectionMatrix(K, pose1))
    P0 = oracle.projection_matrix(K, np.eye(3, 4))
    P1 = oracle.projection_matrix(K, pose)
    tri_o = oracle.triangulate(p0[ok], p1[ok], P0, P1)
    tri = load("tri.bin", np.float64).reshape(-1, 3)
    assert tri.shape == tri_o.shape and np.allclose(tri, tri_o, rtol=1e-6, atol=1e-6)
