"""GPU parity of the default pipeline the reference's driver runs (src/visual_odometry/
visual_odometry.cpp:20-68): detect-when-lost FAST (append semantics) -> pyramidal LK ->
removePoints -> findEssentialMat(LMEDS) -> recoverPose -> pose chaining, through the C++ host
mirror's VisualOdometry class (visual_odometry_amd/host/), against the same state machine restated
here over the CPU oracle.  Points, status-filtered lists and masks bit-exact; E / R / t within 1e-6.
getScale draws from std::mt19937 through libstdc++'s uniform_int_distribution, which numpy cannot
replay: the chained pose is checked with the scale the driver reports."""
import numpy as np
import pytest

from visual_odometry_amd import host_api, synth

pytestmark = pytest.mark.gpu


class _Frame:
    def __init__(self, image):
        self.image = image
        self.points = np.zeros((0, 2), np.float32)
        self.mask = np.zeros(0, np.uint8)
        self.points_3d = np.zeros((0, 3))
        self.pose_R = np.eye(3)
        self.pose_t = np.zeros(3)


def _run(oracle, frames, K, min_tracked, channels, scale_seed=7):
    vo = host_api.HostVisualOdometry(K, min_tracked, scale_seed, channels)
    buf, tracking, detections = [], False, 0
    lk = oracle.lk_params(channels=channels)
    for k, img in enumerate(frames):
        pose = vo.add_image(img)
        st = vo.state()
        buf.append(_Frame(img))
        if len(buf) > 3:
            buf.pop(0)
        if len(buf) < 3:
            assert not st["full"] and np.array_equal(pose, np.eye(3, 4))
            continue
        vo1, vo2 = buf[-2], buf[-1]
        if not tracking:
            kp = oracle.fast_detect(vo1.image, 20, True)
            vo1.points = np.concatenate([vo1.points, np.stack([kp["x"], kp["y"]], 1)]).astype(np.float32)
            tracking = True
            detections += 1
        p1, status = oracle.lk_track(vo1.image, vo2.image, vo1.points, lk)
        vo1.points, vo2.points, _, vo1.points_3d, vo1.mask, _ = oracle.remove_points(
            vo1.points, p1, status, vo1.points_3d, vo1.mask)
        if len(vo2.points) < min_tracked:
            tracking = False
        assert st["full"] and st["tracking"] == tracking, k
        assert np.array_equal(st["prev_points"], vo1.points), k
        assert np.array_equal(st["points"], vo2.points), k
        rc, E, mask, _ = oracle.find_essential(vo1.points, vo2.points, K, oracle.LMEDS)
        assert rc == 0
        ng, R, t, pm, pts4 = oracle.recover_pose(E, vo1.points, vo2.points, K, 200.0, mask)
        vo2.mask = pm
        with np.errstate(divide="ignore", invalid="ignore"):
            vo2.points_3d = (pts4[:3] / pts4[3]).T
        assert np.array_equal(st["mask"], pm), k
        assert min(np.abs(st["E"] - E).max(), np.abs(st["E"] + E).max()) < 1e-6, k
        ok = pm.astype(bool)
        assert np.allclose(st["points_3d"][ok], vo2.points_3d[ok], rtol=1e-6, atol=1e-6), k
        if ng > 8:
            assert np.abs(st["local_R"] - R.T).max() < 1e-6 and np.abs(st["local_t"] + t).max() < 1e-6, k
            s = st["scale"]
            assert 0 < s <= 5
            vo2.pose_t = vo1.pose_t + s * (vo1.pose_R @ (-t))
            vo2.pose_R = R.T @ vo1.pose_R
        else:
            vo2.pose_R, vo2.pose_t = vo1.pose_R, vo1.pose_t
        assert np.abs(pose[:, :3] - vo2.pose_R).max() < 1e-6 and np.abs(pose[:, 3] - vo2.pose_t).max() < 1e-5, k
    vo.close()
    return detections


def _small_sequence(n):
    K = synth.KITTI_K.copy()
    K[0, 0] = K[1, 1] = 718.856 * 640 / 1241.0
    K[0, 2], K[1, 2] = 320.0, 96.0
    frames, poses, _ = synth.make_sequence(5, n, w=640, h=192, K=K)
    return frames, K


def test_driver_default_pipeline(oracle):
    frames, K = _small_sequence(6)
    assert _run(oracle, frames, K, min_tracked=100, channels=3) == 1


def test_driver_redetects_when_tracks_run_out(oracle):
    # a bar no frame can meet: tracking_ drops after every pair, so every addImage detects again
    # and APPENDS to the tracked points (reference: feature_detector.cpp:17-19)
    frames, K = _small_sequence(5)
    assert _run(oracle, frames, K, min_tracked=10 ** 6, channels=1) == 3


def test_sequence_tool_on_a_kitti_layout(tmp_path):
    """BASELINE configs[0] plumbing: KITTI-format directory (PNG frames, calib.txt, poses) -> driver ->
    result poses file + KITTI error metric, through tools/run_sequence.py."""
    import json
    import os
    import subprocess
    import sys

    frames, K = _small_sequence(8)
    _, poses, _ = synth.make_sequence(5, 8, w=16, h=8)  # the ground-truth trajectory of sequence 5
    synth.write_kitti_layout(str(tmp_path), 5, frames, poses, K)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "results"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "run_sequence.py"), "--data_dir", str(tmp_path),
                        "--seq", "05", "--res_dir", str(out), "--min_tracked_features", "100", "--scale_seed", "7"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["frames"] == 8 and rep["pipeline"] == "lk"
    got = host_api.load_poses(str(out / "05.txt"))
    assert got.shape == (8, 3, 4)
    # the same frames through the driver in this process give the same trajectory
    vo = host_api.HostVisualOdometry(K, 100, 7, 3)
    want = np.stack([vo.add_image(f) for f in frames])
    vo.close()
    assert np.allclose(got, want, rtol=1e-8, atol=1e-8)
    assert np.array_equal(got[0], np.eye(3, 4)) and np.array_equal(got[1], np.eye(3, 4))
    assert np.abs(got[-1][:, 3]).max() > 0.5  # it moved
