"""GPU parity of the device-resident batched frame-pair path (vo_pairs_run_dev) against the CPU
oracle run stage by stage: bit-exact keypoints, descriptors and matches; E / R / t within the
north_star tolerance.  This is BASELINE.json config[1] (single frame pair, 2k keypoints)."""
import numpy as np
import pytest
import torch

from visual_odometry_amd import api, synth

pytestmark = pytest.mark.gpu

K = synth.KITTI_K


def rot_angle(Ra, Rb):
    return float(np.linalg.norm(Ra - Rb))


def oracle_pair(oracle, f0, f1, nfeatures):
    po = oracle.orb_params(nfeatures)
    k0, d0 = oracle.orb_detect_compute(f0, po)
    k1, d1 = oracle.orb_detect_compute(f1, po)
    knn, nv = oracle.knn2_hamming(d0, d1)
    good = oracle.match_filter(knn, nv, k0, k1, 0.7, 200.0)
    p0 = np.stack([k0["x"][good["queryIdx"]], k0["y"][good["queryIdx"]]], 1)
    p1 = np.stack([k1["x"][good["trainIdx"]], k1["y"][good["trainIdx"]]], 1)
    rc, E, mask, st = oracle.find_essential(p0, p1, K, oracle.RANSAC)
    ng, R, t, pm, pts = oracle.recover_pose(E, p0, p1, K, 200.0, mask)
    return dict(k0=k0, d0=d0, k1=k1, d1=d1, good=good, E=E, mask=mask, R=R, t=t, ng=ng, pm=pm, rc=rc)


@pytest.mark.parametrize("nfeatures", [2000])
def test_pairs_run_matches_oracle(vo, oracle, frames, nfeatures):
    fr, poses, _ = frames
    d = torch.from_numpy(fr).to("cuda:0")
    torch.cuda.synchronize()
    n, h, w = fr.shape
    res = vo.pairs_run_dev(d.data_ptr(), n, w, h, w, w * h, K, api.pair_params(nfeatures))
    for p in range(n - 1):
        o = oracle_pair(oracle, fr[p], fr[p + 1], nfeatures)
        r = res[p]
        kg, dg = vo.pairs_get_frame(p)
        assert np.array_equal(kg, o["k0"]) and np.array_equal(dg, o["d0"])
        kg1, dg1 = vo.pairs_get_frame(p + 1)
        assert np.array_equal(kg1, o["k1"]) and np.array_equal(dg1, o["d1"])
        good, m1, m2 = vo.pairs_get_matches(p)
        assert np.array_equal(good, o["good"])
        assert r.n_kp0 == len(o["k0"]) and r.n_kp1 == len(o["k1"]) and r.n_good == len(o["good"])
        assert r.status == 0 == o["rc"]
        E = np.array(r.E).reshape(3, 3)
        assert min(np.abs(E - o["E"]).max(), np.abs(E + o["E"]).max()) < 1e-7
        assert np.array_equal(m1, o["mask"]) and r.n_inliers == int(o["mask"].sum())
        assert np.array_equal(m2, o["pm"]) and r.n_pose == o["ng"]
        assert np.abs(np.array(r.R).reshape(3, 3) - o["R"]).max() < 1e-6
        assert np.abs(np.array(r.t) - o["t"]).max() < 1e-6
        # and the pose is the sequence's ground truth (x2 = R x1 + t maps camera p to camera p+1)
        Rgt = poses[p + 1][:, :3].T @ poses[p][:, :3]
        assert np.abs(np.array(r.R).reshape(3, 3) - Rgt).max() < 5e-3


def test_pairs_run_equals_single_calls(vo, frames):
    """The batched path and the host-pointer entry points give identical results."""
    fr, _, _ = frames
    d = torch.from_numpy(fr).to("cuda:0")
    torch.cuda.synchronize()
    n, h, w = fr.shape
    res = vo.pairs_run_dev(d.data_ptr(), n, w, h, w, w * h, K, api.pair_params(1000))
    k0, d0 = vo.pairs_get_frame(0)
    k1, d1 = vo.pairs_get_frame(1)
    good_b, m1_b, _ = vo.pairs_get_matches(0)
    E_b = np.array(res[0].E).reshape(3, 3)
    sk0, sd0 = vo.orb_detect_compute(fr[0], api.orb_params(1000))
    sk1, sd1 = vo.orb_detect_compute(fr[1], api.orb_params(1000))
    assert np.array_equal(k0, sk0) and np.array_equal(d0, sd0) and np.array_equal(d1, sd1)
    knn, nv = vo.knn2_hamming(sd0, sd1)
    good = vo.match_filter(knn, nv, sk0, sk1)
    assert np.array_equal(good, good_b)
    p0 = np.stack([sk0["x"][good["queryIdx"]], sk0["y"][good["queryIdx"]]], 1)
    p1 = np.stack([sk1["x"][good["trainIdx"]], sk1["y"][good["trainIdx"]]], 1)
    rc, E, m = vo.find_essential(p0, p1, K, 8)
    assert np.array_equal(E, E_b) and np.array_equal(m, m1_b)


def test_pairs_run_frames_with_more_corners_than_the_candidate_list():
    """Noise frames (~78 k FAST corners per pyramid at edge threshold 19 and FAST threshold 10) overflow
    the initial candidate lists: the batch goes through the host selection with grown lists and still
    equals the single calls."""
    rng = np.random.default_rng(3)
    base = rng.integers(0, 256, (376, 1241 + 8)).astype(np.float32)
    k = np.ones(3, np.float32) / 3
    base = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 1, base)
    base = np.apply_along_axis(lambda c: np.convolve(c, k, "same"), 0, base)
    fr = np.ascontiguousarray(np.stack([base[:, s:s + 1241] for s in (0, 3, 6)]).astype(np.uint8))  # shifted copies
    ctx = api.VoContext(0)
    try:
        d = torch.from_numpy(fr).to("cuda:0")
        torch.cuda.synchronize()
        prm = api.pair_params(1000)
        prm.orb.edge_threshold = 19
        prm.orb.fast_threshold = 10
        res = ctx.pairs_run_dev(d.data_ptr(), 3, 1241, 376, 1241, 1241 * 376, K, prm)
        assert ctx.get_option(api.OPT_LAST_FLAGS) & 1
        assert ctx.get_option(api.OPT_CAND_CAP) > 65536  # the lists grew
        k0, d0 = ctx.pairs_get_frame(0)
        k1, d1 = ctx.pairs_get_frame(1)
        good_b, _, _ = ctx.pairs_get_matches(0)
        sk0, sd0 = ctx.orb_detect_compute(fr[0], api.orb_params(1000, edge_threshold=19, fast_threshold=10))
        sk1, sd1 = ctx.orb_detect_compute(fr[1], api.orb_params(1000, edge_threshold=19, fast_threshold=10))
        assert len(sk0) > 900
        assert np.array_equal(k0, sk0) and np.array_equal(d0, sd0)
        assert np.array_equal(k1, sk1) and np.array_equal(d1, sd1)
        knn, nv = ctx.knn2_hamming(sd0, sd1)
        good = ctx.match_filter(knn, nv, sk0, sk1)
        assert np.array_equal(good, good_b) and res[0].n_good == len(good)
    finally:
        ctx.close()


def test_pairs_run_with_one_frame_on_the_host_selection_path(vo, frames):
    """Only the middle frame overflows the device selection (a corner-dense noise frame): it alone goes
    through the host selection, its neighbours keep their device picks, and the batch equals the
    single-image calls frame by frame."""
    fr, _, _ = frames
    rng = np.random.default_rng(11)
    noise = rng.integers(0, 256, (376, 1241)).astype(np.float32)
    k = np.ones(3, np.float32) / 3
    noise = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 1, noise)
    noise = np.apply_along_axis(lambda c: np.convolve(c, k, "same"), 0, noise)
    batch = np.ascontiguousarray(np.stack([fr[0], noise.astype(np.uint8), fr[1]]))
    d = torch.from_numpy(batch).to("cuda:0")
    torch.cuda.synchronize()
    res = vo.pairs_run_dev(d.data_ptr(), 3, 1241, 376, 1241, 1241 * 376, K, api.pair_params(1000))
    assert vo.get_option(api.OPT_LAST_FLAGS) & 1
    got = [vo.pairs_get_frame(f) for f in range(3)]
    for f in range(3):
        sk, sd = vo.orb_detect_compute(batch[f], api.orb_params(1000))
        assert len(sk) > 900
        assert np.array_equal(got[f][0], sk) and np.array_equal(got[f][1], sd), f
    assert res[0].n_kp0 == len(got[0][0]) and res[1].n_kp1 == len(got[2][0])


def test_pairs_run_flat_frames(vo):
    """No keypoints anywhere: every pair reports status VO_ERR_FAILED and zero counts."""
    fr = np.full((3, 376, 1241), 77, np.uint8)
    d = torch.from_numpy(fr).to("cuda:0")
    torch.cuda.synchronize()
    res = vo.pairs_run_dev(d.data_ptr(), 3, 1241, 376, 1241, 1241 * 376, K, api.pair_params(2000))
    for p in range(2):
        assert res[p].n_kp0 == 0 and res[p].n_good == 0 and res[p].status == 4


def test_pairs_run_cross_check_option(vo, frames):
    """cross_check = 1 in vo_pair_params: the batched path keeps exactly the matches the single calls
    keep with vo_knn2_hamming_cross + vo_match_filter_cross, and a subset of the default's."""
    fr, _, _ = frames
    d = torch.from_numpy(fr).to("cuda:0")
    torch.cuda.synchronize()
    n, h, w = fr.shape
    vo.pairs_run_dev(d.data_ptr(), n, w, h, w, w * h, K, api.pair_params(1000))
    plain, _, _ = vo.pairs_get_matches(0)
    res = vo.pairs_run_dev(d.data_ptr(), n, w, h, w, w * h, K, api.pair_params(1000, cross_check=1))
    k0, d0 = vo.pairs_get_frame(0)
    k1, d1 = vo.pairs_get_frame(1)
    good_b, m1_b, _ = vo.pairs_get_matches(0)
    knn, nv, rev = vo.knn2_hamming_cross(d0, d1)
    good = vo.match_filter(knn, nv, k0, k1, rev_nn=rev)
    assert np.array_equal(good, good_b) and res[0].n_good == len(good)
    assert 0 < len(good) < len(plain)
    assert set(map(tuple, good[["queryIdx", "trainIdx"]].tolist())) <= set(map(tuple, plain[["queryIdx", "trainIdx"]].tolist()))
    assert res[0].status == 0 and res[0].n_inliers == int(m1_b.sum())


def test_pairs_run_stress_config_end_to_end(oracle):
    """BASELINE configs[2] whole, not by stage: nfeatures 8192, 8 levels, 4096 RANSAC hypotheses through
    vo_pairs_run_dev against the oracle run stage by stage on the same two frames."""
    fr, _, _ = synth.make_sequence(21, 2)
    ctx = api.VoContext(0)
    try:
        d = torch.from_numpy(fr).to("cuda:0")
        torch.cuda.synchronize()
        n, h, w = fr.shape
        res = ctx.pairs_run_dev(d.data_ptr(), n, w, h, w, w * h, K, api.pair_params(8192, max_iters=4096))
        po = oracle.orb_params(8192)
        k0, d0 = oracle.orb_detect_compute(fr[0], po)
        k1, d1 = oracle.orb_detect_compute(fr[1], po)
        assert len(k0) > 7000
        g0 = ctx.pairs_get_frame(0)
        g1 = ctx.pairs_get_frame(1)
        assert np.array_equal(g0[0], k0) and np.array_equal(g0[1], d0)
        assert np.array_equal(g1[0], k1) and np.array_equal(g1[1], d1)
        knn, nv = oracle.knn2_hamming(d0, d1)
        good = oracle.match_filter(knn, nv, k0, k1, 0.7, 200.0)
        gg, m1, m2 = ctx.pairs_get_matches(0)
        assert np.array_equal(gg, good)
        p0 = np.stack([k0["x"][good["queryIdx"]], k0["y"][good["queryIdx"]]], 1)
        p1 = np.stack([k1["x"][good["trainIdx"]], k1["y"][good["trainIdx"]]], 1)
        rc, E, mask, _ = oracle.find_essential(p0, p1, K, oracle.RANSAC, max_iters=4096)
        r = res[0]
        assert rc == r.status == 0
        Eg = np.array(r.E).reshape(3, 3)
        # north_star's tolerance (1e-6): the winning sample of this pair is ill-conditioned enough that the two
        # solvers (QR + Sturm here, Jacobi SVD + Durand-Kerner in the oracle) differ by 1.1e-7 in E
        assert min(np.abs(Eg - E).max(), np.abs(Eg + E).max()) < 1e-6
        assert np.array_equal(m1, mask) and r.n_inliers == int(mask.sum())
        ng, R, t, pm, _ = oracle.recover_pose(E, p0, p1, K, 200.0, mask)
        assert np.array_equal(m2, pm) and r.n_pose == ng
        assert np.abs(np.array(r.R).reshape(3, 3) - R).max() < 1e-6 and np.abs(np.array(r.t) - t).max() < 1e-6
    finally:
        ctx.close()
