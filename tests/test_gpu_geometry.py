"""GPU parity: findEssentialMat (RANSAC and LMedS), recoverPose, triangulate through the C ABI
against the CPU oracle, same cv::RNG seed.  Floating point (fp64) bar from BASELINE.json's
north_star: rotation <= 1e-6 rad, unit translation <= 1e-6; inlier masks are compared exactly.
Mirrors the calls in matcher() (src/matcher/matcher.cpp:42) and updatePose()
(src/visual_odometry/vo_pose.h:14-22)."""
import numpy as np
import pytest

from visual_odometry_amd import synth

pytestmark = pytest.mark.gpu

K = synth.KITTI_K
ROT_TOL = 1e-6   # rad
T_TOL = 1e-6     # unit translation
E_TOL = 1e-7     # unit-Frobenius E, up to sign


def rot_angle(Ra, Rb):
    c = (np.trace(Ra.T @ Rb) - 1) / 2
    return float(np.arccos(np.clip(c, -1, 1))) if abs(c) < 1 - 1e-13 else float(np.linalg.norm(Ra - Rb))


def e_diff(Ea, Eb):
    return min(np.abs(Ea - Eb).max(), np.abs(Ea + Eb).max())


@pytest.mark.parametrize("n,seed,ratio", [(500, 0, 0.6), (2000, 1, 0.6), (1000, 2, 0.85), (300, 3, 0.4),
                                          (64, 4, 0.7), (2000, 5, 0.25)])
def test_find_essential_ransac(vo, oracle, n, seed, ratio):
    p0, p1, R, t, inl = synth.correspondences(n, seed=seed, inlier_ratio=ratio)
    rc_o, E_o, m_o, st = oracle.find_essential(p0, p1, K, oracle.RANSAC)
    rc_g, E_g, m_g = vo.find_essential(p0, p1, K, 8)
    assert rc_g == rc_o == 0
    assert e_diff(E_g, E_o) < E_TOL
    assert np.array_equal(m_g, m_o)
    assert m_o.sum() == st[2]


@pytest.mark.parametrize("n,seed,ratio", [(750, 10, 0.8), (2000, 11, 0.7), (301, 12, 0.9), (6000, 13, 0.75)])
def test_find_essential_lmeds(vo, oracle, n, seed, ratio):
    # updatePose(): LMEDS on tracked points, n = 750..6000, odd and even n
    p0, p1, R, t, inl = synth.correspondences(n, seed=seed, inlier_ratio=ratio)
    rc_o, E_o, m_o, st = oracle.find_essential(p0, p1, K, oracle.LMEDS)
    rc_g, E_g, m_g = vo.find_essential(p0, p1, K, 4)
    assert st[0] == 134  # fixed LMedS iteration count
    assert rc_g == rc_o == 0
    assert e_diff(E_g, E_o) < E_TOL
    assert np.array_equal(m_g, m_o)


def test_find_essential_stress_4096_hypotheses(vo, oracle):
    # BASELINE config 3: 4096 hypotheses, low inlier ratio so the cap does not shrink early
    p0, p1, R, t, inl = synth.correspondences(4000, seed=21, inlier_ratio=0.12)
    rc_o, E_o, m_o, st = oracle.find_essential(p0, p1, K, oracle.RANSAC, max_iters=4096)
    rc_g, E_g, m_g = vo.find_essential(p0, p1, K, 8, max_iters=4096)
    assert st[0] > 1000  # more iterations than cv 3.4.2's cap were actually needed
    assert rc_g == rc_o == 0
    assert e_diff(E_g, E_o) < E_TOL
    assert np.array_equal(m_g, m_o)


def test_find_essential_recovers_ground_truth(vo):
    p0, p1, R, t, inl = synth.correspondences(1500, seed=31, inlier_ratio=0.7, sigma=0.0)
    rc, E, m = vo.find_essential(p0, p1, K, 8)
    ng, Rg, tg, m2, pts = vo.recover_pose(E, p0, p1, K, 200.0, m)
    assert rot_angle(Rg, R) < 2e-3
    assert np.linalg.norm(tg - t) < 5e-3  # float32 pixel coordinates limit the accuracy
    assert ng > 0.6 * 1500


@pytest.mark.parametrize("n", [0, 3, 4])
def test_find_essential_too_few_points(vo, oracle, n):
    p0, p1, *_ = synth.correspondences(10, seed=1)
    rc_g, E_g, m_g = vo.find_essential(p0[:n], p1[:n], K, 8)
    rc_o, E_o, m_o, _ = oracle.find_essential(p0[:n], p1[:n], K, oracle.RANSAC)
    assert rc_g == rc_o == 4  # cv returns an empty matrix
    assert not E_g.any() and not m_g.any()


def test_find_essential_exactly_five_points(vo, oracle):
    p0, p1, *_ = synth.correspondences(5, seed=2, inlier_ratio=1.0, sigma=0.0)
    for method in (8, 4):
        rc_g, E_g, m_g = vo.find_essential(p0, p1, K, method)
        assert rc_g == 0 and m_g.all()
        # E is one of the exact solutions of the five points
        x0 = np.c_[(p0[:, 0] - K[0, 2]) / K[0, 0], (p0[:, 1] - K[1, 2]) / K[1, 1], np.ones(5)]
        x1 = np.c_[(p1[:, 0] - K[0, 2]) / K[0, 0], (p1[:, 1] - K[1, 2]) / K[1, 1], np.ones(5)]
        assert np.abs(np.einsum("ni,ij,nj->n", x1, E_g, x0)).max() < 1e-9


def test_find_essential_seed_changes_samples(vo, oracle):
    p0, p1, *_ = synth.correspondences(800, seed=6, inlier_ratio=0.5)
    for seed in (2**64 - 1, 12345):
        rc_o, E_o, m_o, _ = oracle.find_essential(p0, p1, K, oracle.RANSAC, seed=seed)
        rc_g, E_g, m_g = vo.find_essential(p0, p1, K, 8, seed=seed)
        assert e_diff(E_g, E_o) < E_TOL and np.array_equal(m_g, m_o)


@pytest.mark.parametrize("n,seed", [(500, 40), (2000, 41), (1, 42)])
def test_recover_pose(vo, oracle, n, seed):
    p0, p1, R, t, inl = synth.correspondences(max(n, 100), seed=seed, inlier_ratio=0.7)
    rc, E, mask, _ = oracle.find_essential(p0, p1, K, oracle.RANSAC)
    p0, p1, mask = p0[:n], p1[:n], mask[:n]
    for use_mask in (True, False):
        mi = mask if use_mask else None
        ng_o, R_o, t_o, m_o, q_o = oracle.recover_pose(E, p0, p1, K, 200.0, mi)
        ng_g, R_g, t_g, m_g, q_g = vo.recover_pose(E, p0, p1, K, 200.0, mi)
        assert ng_g == ng_o
        assert rot_angle(R_g, R_o) < ROT_TOL
        assert np.linalg.norm(t_g - t_o) < T_TOL
        if use_mask:
            assert np.array_equal(m_g, m_o)
        # points4DToVec: dehomogenised points agree (homogeneous sign/scale is arbitrary)
        w_o, w_g = q_o[3], q_g[3]
        ok = (np.abs(w_o) > 1e-9) & (np.abs(w_g) > 1e-9)
        X_o, X_g = q_o[:3, ok] / w_o[ok], q_g[:3, ok] / w_g[ok]
        assert np.allclose(X_g, X_o, rtol=1e-6, atol=1e-6)


def test_recover_pose_empty(vo, oracle):
    E = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 0.0]])
    ng, R, t, m, q = vo.recover_pose(E, np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32), K)
    assert ng == 0 and abs(np.linalg.det(R) - 1) < 1e-9


def test_triangulate(vo, oracle):
    # triangulate(points0, points1, P0, P1) with projection matrices from getProjectionMatrix
    rng = np.random.default_rng(7)
    pose0 = np.hstack([np.eye(3), np.zeros((3, 1))])
    pose1 = np.hstack([synth.rotation([0.01, -0.03, 0.005]), np.array([[0.1], [-0.02], [1.0]])])
    P0g, P1g = vo.projection_matrix(K, pose0), vo.projection_matrix(K, pose1)
    assert np.array_equal(P0g, oracle.projection_matrix(K, pose0))
    assert np.array_equal(P1g, oracle.projection_matrix(K, pose1))
    X = rng.uniform([-8, -2, 5], [8, 2, 50], (700, 3))
    Xh = np.c_[X, np.ones(len(X))]
    a = (P0g @ Xh.T).T
    b = (P1g @ Xh.T).T
    p0 = (a[:, :2] / a[:, 2:]).astype(np.float32)
    p1 = (b[:, :2] / b[:, 2:]).astype(np.float32)
    g = vo.triangulate(p0, p1, P0g, P1g)
    o = oracle.triangulate(p0, p1, P0g, P1g)
    assert np.allclose(g, o, rtol=1e-7, atol=1e-7)
    assert np.median(np.linalg.norm(g - X, axis=1)) < 0.05
    assert vo.triangulate(p0[:0], p1[:0], P0g, P1g).shape == (0, 3)


def _five_point_problems(count, seed):
    """A third real two-view geometries with noise, the rest unrelated points (what a RANSAC sample with an
    outlier looks like), as tests/test_host_solver.py draws them."""
    rng = np.random.default_rng(seed)
    a = rng.uniform(-0.6, 0.6, (count, 5, 2))
    b = rng.uniform(-0.6, 0.6, (count, 5, 2))
    for i in range(0, count, 3):
        X = rng.uniform([-8, -3, 4], [8, 3, 40], (5, 3))
        R = synth.rotation(rng.normal(0, 0.05, 3))
        t = rng.normal(0, 1, 3)
        Y = X @ R.T + t / np.linalg.norm(t)
        a[i] = X[:, :2] / X[:, 2:] + rng.normal(0, 1e-3, (5, 2))
        b[i] = Y[:, :2] / Y[:, 2:] + rng.normal(0, 1e-3, (5, 2))
    return a, b


@pytest.mark.gpu
def test_five_point_lane_and_group_solvers_agree(vo, oracle):
    """vo_five_point variant 0 (a problem per lane) and 1 (a problem per 16-lane group) are two schedules of one
    algorithm: same number of models, same order (ascending root), same matrices up to the conditioning of the
    root; both match the oracle's runKernel on a sample of the problems.  Tolerances: typical agreement is
    1e-13; nearly double roots move by ~1e-5, and a pair of roots closer than the solvers can separate may be
    found by one and skipped by the other (allowed on at most 0.2 % of the problems)."""
    count = 6000
    a, b = _five_point_problems(count, 11)
    m0, n0 = vo.five_point(a, b, variant=0)
    m1, n1 = vo.five_point(a, b, variant=1)
    assert n0.min() >= 0 and n0.max() <= 10 and n1.max() <= 10
    same = n0 == n1
    assert (~same).sum() <= count * 0.002, f"{(~same).sum()} problems differ in the number of models"
    assert n1.sum() > count  # the typical problem has 2-6 real solutions
    diffs = []
    for i in np.nonzero(same)[0]:
        if n0[i]:
            diffs.append(np.abs(m0[i, :n0[i]] - m1[i, :n1[i]]).max())
    diffs = np.array(diffs)
    assert np.median(diffs) < 1e-12 and np.percentile(diffs, 99) < 1e-7 and diffs.max() < 1e-3, (
        np.median(diffs), np.percentile(diffs, 99), diffs.max())
    # every model of the group solver satisfies the five epipolar constraints
    for i in range(0, count, 7):
        for E in m1[i, :n1[i]]:
            r = [np.r_[q, 1] @ E @ np.r_[p, 1] for p, q in zip(a[i], b[i])]
            assert np.abs(r).max() < 1e-8
            assert abs(np.linalg.norm(E) - 1) < 1e-12
    # unused slots stay zero
    for i in range(0, count, 50):
        assert not m1[i, n1[i]:].any()
    # against the oracle (cv's runKernel restated): count and matrices
    od = []
    for i in range(0, count, 20):
        Eo = oracle.five_point(np.ascontiguousarray(a[i]), np.ascontiguousarray(b[i]))
        if len(Eo) != n1[i]:
            continue
        for E in m1[i, :n1[i]]:
            od.append(min(np.abs(E - F).max() for F in Eo))
    od = np.array(od)
    assert len(od) > 500 and np.median(od) < 1e-12 and np.percentile(od, 99) < 1e-6


@pytest.mark.gpu
def test_five_point_degenerate_inputs(vo):
    """Inputs the solvers must survive without a fault or a hang: repeated points, all-zero points, collinear
    points, huge coordinates.  Whatever they return must be finite and agree in count between the variants
    (or be 0 models)."""
    a = np.zeros((8, 5, 2))
    b = np.zeros((8, 5, 2))
    rng = np.random.default_rng(3)
    a[1] = b[1] = rng.uniform(-0.5, 0.5, (5, 2))          # identical views: pure rotation by identity
    a[2, :] = rng.uniform(-0.5, 0.5, 2)                    # one point five times
    b[2, :] = rng.uniform(-0.5, 0.5, 2)
    a[3, :, 0] = np.linspace(-0.5, 0.5, 5)                 # collinear in both views
    b[3, :, 0] = np.linspace(-0.4, 0.6, 5)
    a[4] = rng.uniform(-1e6, 1e6, (5, 2))
    b[4] = rng.uniform(-1e6, 1e6, (5, 2))
    a[5] = rng.uniform(-1e-9, 1e-9, (5, 2))
    b[5] = rng.uniform(-1e-9, 1e-9, (5, 2))
    a[6], b[6] = _five_point_problems(1, 5)
    a[7], b[7] = a[6], b[6]
    for variant in (0, 1):
        m, n = vo.five_point(a, b, variant=variant)
        assert n.min() >= 0 and n.max() <= 10
        assert np.isfinite(m).all()
        assert n[6] == n[7] and np.array_equal(m[6], m[7])  # deterministic


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["RANSAC", "LMEDS"])
def test_essential_pose_equals_the_two_calls(vo, method):
    """vo_essential_pose is vo_find_essential + vo_recover_pose (mask passed through) in one submission: every
    output identical, including the no-model case."""
    from visual_odometry_amd import api
    meth = getattr(api, method)
    for seed, n, ratio in ((1, 800, 0.7), (2, 3000, 0.5), (3, 60, 0.9), (4, 5, 1.0)):
        p0, p1, *_ = synth.correspondences(n, seed=seed, inlier_ratio=ratio)
        rc, E, m = vo.find_essential(p0, p1, K, method=meth)
        rc2, E2, ng2, R2, t2, m2, pts2 = vo.essential_pose(p0, p1, K, method=meth)
        assert rc == rc2
        assert np.array_equal(E, E2)
        if rc == 0:
            ng, R, t, mp, pts = vo.recover_pose(E, p0, p1, K, 200.0, m)
            assert ng == ng2 and np.array_equal(R, R2) and np.array_equal(t, t2)
            assert np.array_equal(mp, m2) and np.array_equal(pts, pts2)
    rc2, E2, ng2, R2, t2, m2, pts2 = vo.essential_pose(p0[:3], p1[:3], K, method=meth)
    assert rc2 != 0 and ng2 == 0 and not E2.any() and not m2.any()


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["lmeds", "ransac"])
def test_essential_pose_batch_equals_single_calls(vo, method):
    """vo_essential_pose_batch: S point sets in one submission give, set for set, the bits of vo_essential_pose -
    sets of different sizes, one with fewer than five points (cv's empty-matrix case), one of pure noise."""
    from visual_odometry_amd import api

    m = api.LMEDS if method == "lmeds" else api.RANSAC
    sets = []
    for k, (n, ratio) in enumerate([(900, 0.8), (2500, 0.6), (3, 1.0), (400, 0.9), (64, 0.0), (5, 1.0), (1300, 0.7)]):
        p0, p1, *_ = synth.correspondences(max(n, 8), seed=20 + k, inlier_ratio=ratio)
        sets.append((p0[:n], p1[:n]))
    want = [vo.essential_pose(a, b, K, method=m) for a, b in sets]
    got = vo.essential_pose_batch([a for a, _ in sets], [b for _, b in sets], K, method=m)
    assert want[2][0] == api.VO_ERR_FAILED and want[0][0] == 0 and want[1][0] == 0
    for w, g in zip(want, got):
        assert w[0] == g[0] and w[2] == g[2]
        for i in (1, 3, 4, 5, 6):
            assert np.array_equal(w[i], g[i]), i
