"""CPU tests of the host mirror's file formats and bookkeeping (visual_odometry_amd/host/, no GPU
work): loadKittiCalibration / loadPoses / savePoses on a KITTI-layout directory written by the
synthetic generator (reference: src/kitti/kitti.cpp:10-49, README.md:54-94) and removePoints()
against the oracle's restatement (reference: src/features/utils.h:9-58)."""
import ctypes as C
import os

import numpy as np
import pytest

from visual_odometry_amd import host_api, synth


def test_host_library_exports():
    lib = host_api.lib()
    for s in host_api.SYMBOLS:
        assert hasattr(lib, s), s


def test_kitti_layout_round_trip(tmp_path):
    frames = np.zeros((3, 8, 8), np.uint8)
    poses = np.stack([np.hstack([synth.rotation(np.array([0.0, 0.01 * k, 0.0])), [[0.1 * k], [0.0], [1.0 * k]]])
                      for k in range(3)])
    synth.write_kitti_layout(str(tmp_path), 7, frames, poses)
    seq = tmp_path / "sequences" / "07"
    K = host_api.load_calibration(str(seq / "calib.txt"), 0)
    assert np.array_equal(K, synth.KITTI_K)
    assert np.array_equal(K, synth.load_kitti_calibration(str(seq / "calib.txt"), 0))
    assert np.array_equal(host_api.load_calibration(str(seq / "calib.txt"), 3), K)
    got = host_api.load_poses(str(tmp_path / "poses" / "07.txt"))
    assert got.shape == (3, 3, 4) and np.allclose(got, poses, rtol=0, atol=1e-6)
    out = tmp_path / "res.txt"
    host_api.save_poses(str(out), got)
    assert np.allclose(host_api.load_poses(str(out)), got, rtol=1e-9, atol=1e-12)
    with pytest.raises(RuntimeError):
        host_api.load_calibration(str(seq / "missing.txt"), 0)
    assert len(host_api.load_poses(str(tmp_path / "poses" / "missing.txt"))) == 0


def test_calibration_line_is_parsed_like_the_reference(tmp_path):
    # "P0: " is dropped by erase(0, 4); the rest is split on single spaces into 12 doubles
    p = tmp_path / "calib.txt"
    p.write_text("P0: 7.188560000000e+02 0 6.071928000000e+02 0 0 7.188560000000e+02 1.852157000000e+02 0 0 0 1 0\n"
                 "P1: 1 2 3 4 5 6 7 8 9 10 11 12\n")
    K = host_api.load_calibration(str(p), 0)
    assert K[0, 0] == 718.856 and K[0, 2] == 607.1928 and K[1, 2] == 185.2157 and K[2, 2] == 1
    assert np.array_equal(host_api.load_calibration(str(p), 1), [[1, 2, 3], [5, 6, 7], [9, 10, 11]])
    # a double space makes an empty token: std::stod throws in the reference, an error here
    p.write_text("P0: 1  2 3 4 5 6 7 8 9 10 11 12\n")
    with pytest.raises(RuntimeError):
        host_api.load_calibration(str(p), 0)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_remove_points_equals_oracle(oracle, seed):
    rng = np.random.default_rng(seed)
    n = 300 + 50 * seed
    p0 = (rng.random((n, 2)) * 600).astype(np.float32)
    p1 = p0 + rng.normal(0, 3, (n, 2)).astype(np.float32)
    p1[rng.integers(0, n, 5)] -= 700.0          # negative coordinates
    p1[rng.integers(0, n, 5)] += 90.0           # displacement outliers
    status = (rng.random(n) > 0.15).astype(np.uint8)
    n3 = [0, n, n - 40][seed]
    nm = [n + 10, n, 0][seed]
    pts3d = rng.random((n3, 3))
    mask = rng.integers(0, 2, nm).astype(np.uint8)
    a = host_api.remove_points(p0, p1, status, pts3d, mask)
    b = oracle.remove_points(p0, p1, status, pts3d, mask)
    for x, y in zip(a[:5], b[:5]):
        assert np.array_equal(x, y)
    assert a[5] == b[5]


def test_kitti_metric_on_known_drift():
    # straight line, 1 m per frame; the estimate is scaled by 1.02 -> 2 % of the distance actually covered, no
    # rotation error.  A sub-sequence ends at the first frame whose distance is strictly greater than
    # start + length (the devkit's rule: reference src/kitti/evaluate_odometry.cpp, lastFrameFromSegmentLength),
    # so with exact hits at 1 m spacing it covers length + 1 metres.
    n = 1000
    gt = np.tile(np.eye(3, 4), (n, 1, 1))
    gt[:, 2, 3] = np.arange(n)
    res = gt.copy()
    res[:, 2, 3] *= 1.02
    t, r, cnt = host_api.kitti_sequence_errors(gt, res)
    want, k = 0.0, 0
    for first in range(0, n, 10):
        for L in (100, 200, 300, 400, 500, 600, 700, 800):
            last = next((i for i in range(first, n) if gt[i, 2, 3] > gt[first, 2, 3] + L), -1)
            if last < 0:
                continue
            assert last - first == L + 1
            want += 0.02 * (L + 1) / L
            k += 1
    assert cnt == k and abs(t - want / k) < 1e-9 and r < 1e-7
    assert host_api.kitti_sequence_errors(gt[:50], res[:50]) is None  # shorter than 100 m


def test_load_poses_tolerates_blank_lines_and_stops_at_garbage(tmp_path):
    p = tmp_path / "p.txt"
    rows = [" ".join(str(float(i * 12 + j)) for j in range(12)) for i in range(3)]
    p.write_text(rows[0] + "\n\n" + rows[1] + "\n" + "1 2 3\n" + rows[2] + "\n")
    got = host_api.load_poses(str(p))
    assert got.shape == (2, 3, 4) and got[1, 2, 3] == 23.0
