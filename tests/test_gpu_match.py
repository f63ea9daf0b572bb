"""GPU parity: brute-force Hamming 2-NN and the ratio/displacement filter, through the C ABI,
against the CPU oracle.  Bit-exact bar: indices, distances, order.
Reads like the reference's use in matcher() (src/matcher/matcher.cpp:17-38)."""
import numpy as np
import pytest

from visual_odometry_amd import api, synth

pytestmark = pytest.mark.gpu


def _same_knn(a, b):
    assert a.shape == b.shape
    for f in ("queryIdx", "trainIdx", "imgIdx", "distance"):
        assert np.array_equal(a[f], b[f]), f


@pytest.mark.parametrize("nq,nt", [(1, 2), (7, 5), (64, 64), (257, 1000), (2000, 2000), (2000, 1999), (3001, 777)])
def test_knn2_random(vo, oracle, nq, nt):
    rng = np.random.default_rng(nq * 7919 + nt)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    g, gv = vo.knn2_hamming(q, t)
    o, ov = oracle.knn2_hamming(q, t)
    _same_knn(g, o)
    assert np.array_equal(gv, ov)


def test_knn2_planted_2k(vo, oracle):
    q, t = synth.matcher_descriptors(2000)
    g, gv = vo.knn2_hamming(q, t)
    o, ov = oracle.knn2_hamming(q, t)
    _same_knn(g, o)


def test_knn2_ties_keep_lower_train_index(vo, oracle):
    # few distinct rows, many exact duplicates -> every query has tied neighbours
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, (6, 32), dtype=np.uint8)
    q = base[rng.integers(0, 6, 300)]
    t = base[rng.integers(0, 6, 500)]
    g, _ = vo.knn2_hamming(q, t)
    o, _ = oracle.knn2_hamming(q, t)
    _same_knn(g, o)
    assert (g["distance"][:, 0] == 0).all() and (g["trainIdx"][:, 0] < g["trainIdx"][:, 1]).all()


@pytest.mark.parametrize("nt", [0, 1])
def test_knn2_fewer_than_two_train_rows(vo, oracle, nt):
    rng = np.random.default_rng(1)
    q = rng.integers(0, 256, (10, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    g, gv = vo.knn2_hamming(q, t)
    o, ov = oracle.knn2_hamming(q, t)
    assert np.array_equal(gv, ov) and (gv == nt).all()
    assert np.array_equal(g["trainIdx"], o["trainIdx"])
    # the reference indexes matches[k][1] unconditionally; such rows are skipped by the filter
    kq = np.zeros(10, oracle.KP_DTYPE)
    kt = np.zeros(max(nt, 1), oracle.KP_DTYPE)
    assert len(vo.match_filter(g, gv, kq, kt)) == 0
    assert len(oracle.match_filter(o, ov, kq, kt)) == 0


def test_knn2_empty_query(vo):
    g, gv = vo.knn2_hamming(np.zeros((0, 32), np.uint8), np.zeros((5, 32), np.uint8))
    assert g.shape == (0, 2)


@pytest.mark.parametrize("d", [61, 64, 16])
def test_knn2_other_descriptor_sizes(vo, oracle, d):
    # AKAZE's MLDB descriptors are 61 bytes (what addKeyFrame feeds matcher(),
    # src/sfm/bundle_adjustment.cpp:41,58)
    rng = np.random.default_rng(d)
    q = rng.integers(0, 256, (300, d), dtype=np.uint8)
    t = rng.integers(0, 256, (400, d), dtype=np.uint8)
    g, _ = vo.knn2_hamming(q, t)
    o, _ = oracle.knn2_hamming(q, t)
    _same_knn(g, o)


def test_knn2_stress_8k(vo, oracle):
    # BASELINE config 3 shape; the oracle takes a few seconds
    q, t = synth.matcher_descriptors(8192)
    g, _ = vo.knn2_hamming(q, t)
    o, _ = oracle.knn2_hamming(q, t)
    _same_knn(g, o)


def test_match_filter_parity(vo, oracle):
    rng = np.random.default_rng(3)
    n = 2000
    q, t = synth.matcher_descriptors(n)
    kq = np.zeros(n, oracle.KP_DTYPE)
    kt = np.zeros(n, oracle.KP_DTYPE)
    kq["x"] = rng.uniform(0, 1241, n).astype(np.float32)
    kq["y"] = rng.uniform(0, 376, n).astype(np.float32)
    kt["x"] = rng.uniform(0, 1241, n).astype(np.float32)
    kt["y"] = rng.uniform(0, 376, n).astype(np.float32)
    o, ov = oracle.knn2_hamming(q, t)
    # make displacement matter: planted matches sit near their query, with some exactly at the gate
    perm = o["trainIdx"][:, 0]
    kt["x"][perm] = kq["x"] + rng.normal(0, 120, n).astype(np.float32)
    kt["y"][perm] = kq["y"] + rng.normal(0, 120, n).astype(np.float32)
    kt["x"][perm[:5]] = kq["x"][:5] + np.float32(200.0)
    kt["y"][perm[:5]] = kq["y"][:5]
    g, gv = vo.knn2_hamming(q, t)
    good_g = vo.match_filter(g, gv, kq, kt, 0.7, 200.0)
    good_o = oracle.match_filter(o, ov, kq, kt, 0.7, 200.0)
    assert 100 < len(good_o) < n
    assert np.array_equal(good_g, good_o)


def test_match_filter_ratio_boundary(vo, oracle):
    # d0 < 0.7f * d1 evaluated in float: 7 < 0.7f*10 is true in float (0.7f*10 = 7.0000002...)
    knn = np.zeros((3, 2), oracle.DM_DTYPE)
    knn["queryIdx"][:, 0] = knn["queryIdx"][:, 1] = np.arange(3)
    knn["trainIdx"][:, 0] = [0, 1, 2]
    knn["trainIdx"][:, 1] = [1, 2, 0]
    knn["distance"][:, 0] = [7, 14, 70]
    knn["distance"][:, 1] = [10, 20, 100]
    kq = np.zeros(3, oracle.KP_DTYPE)
    kt = np.zeros(3, oracle.KP_DTYPE)
    nv = np.full(3, 2, np.int32)
    assert np.array_equal(vo.match_filter(knn, nv, kq, kt), oracle.match_filter(knn, nv, kq, kt))


def test_knn2_complementary_64_byte_rows(vo, oracle):
    # distance 512 = 2^9 needs the tenth key bit: q = ~t at 64 bytes must come out LAST, not first
    rng = np.random.default_rng(11)
    t = rng.integers(0, 256, (70, 64), dtype=np.uint8)
    q = t[:40].copy()
    q[::2] = ~q[::2]
    g, _ = vo.knn2_hamming(q, t)
    o, _ = oracle.knn2_hamming(q, t)
    _same_knn(g, o)
    assert (g["distance"][1::2, 0] == 0).all() and (g["distance"][::2, 0] > 0).all()
    t2 = np.stack([t[0], ~t[0]])
    g2, _ = vo.knn2_hamming(t[:1], t2)
    assert list(g2["distance"][0]) == [0.0, 512.0] and list(g2["trainIdx"][0]) == [0, 1]


def _rev_nn_numpy(q, t):
    """Nearest query of every train row: Hamming distance, ties to the lower query index."""
    lut = np.array([bin(i).count("1") for i in range(256)], np.int32)
    out = np.full(len(t), -1, np.int32)
    for j in range(len(t)):
        if len(q):
            out[j] = int(np.argmin(lut[np.bitwise_xor(q, t[j][None, :])].sum(1)))  # argmin: first minimum
    return out


@pytest.mark.parametrize("nq,nt", [(300, 500), (2000, 1999), (130, 64), (5, 700)])
def test_cross_check_option(vo, oracle, nq, nt):
    """north_star's cross-check filter (off in the reference, matcher.cpp:10): rule stated in
    include/vo_hip.h, restated here in numpy on top of the oracle's 2-NN + ratio/gate filter."""
    rng = np.random.default_rng(nq + nt)
    q, t = synth.matcher_descriptors(max(nq, nt), seed_q=nq, seed_t=nt)
    q, t = q[:nq], t[:nt]
    t[::7] = t[0]  # duplicates: ties in both directions
    kq = np.zeros(nq, oracle.KP_DTYPE)
    kt = np.zeros(nt, oracle.KP_DTYPE)
    for k in (kq, kt):
        k["x"] = rng.uniform(0, 300, len(k)).astype(np.float32)
        k["y"] = rng.uniform(0, 100, len(k)).astype(np.float32)
    g, gv, rev = vo.knn2_hamming_cross(q, t)
    o, ov = oracle.knn2_hamming(q, t)
    _same_knn(g, o)
    assert np.array_equal(gv, ov)
    assert np.array_equal(rev, _rev_nn_numpy(q, t))
    plain = oracle.match_filter(o, ov, kq, kt, 0.9, 200.0)
    want = plain[rev[plain["trainIdx"]] == plain["queryIdx"]]
    got = vo.match_filter(g, gv, kq, kt, 0.9, 200.0, rev_nn=rev)
    assert np.array_equal(got, want)
    assert 0 < len(want) <= len(plain)
    assert np.array_equal(vo.match_filter(g, gv, kq, kt, 0.9, 200.0), plain)  # default stays the reference's


@pytest.mark.parametrize("nq,nt", [(1, 1), (31, 33), (64, 64), (65, 129), (500, 257), (2000, 1987), (3000, 4100)])
def test_knn2_matrix_core_kernel_equals_popcount_kernel(vo, oracle, nq, nt):
    """32-byte rows: k_knn2_mfma (int8 dot products on the matrix cores, the default) and k_knn2<8> (xor +
    popcount, VO_OPT_KNN_MFMA = 0) are two independent formulations of the same integers; both must give the oracle's
    table, ties included (rows drawn from 8 distinct values make most distances tie)."""
    rng = np.random.default_rng(nq * 7919 + nt)
    for kind in ("random", "ties"):
        if kind == "random":
            q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
            t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
        else:
            pool = rng.integers(0, 256, (8, 32), dtype=np.uint8)
            q = pool[rng.integers(0, 8, nq)]
            t = pool[rng.integers(0, 8, nt)]
        vo.set_option(api.OPT_KNN_MFMA, 0)
        try:
            gp, vp = vo.knn2_hamming(q, t)
        finally:
            vo.set_option(api.OPT_KNN_MFMA, 1)
        gm, vm = vo.knn2_hamming(q, t)
        o, ov = oracle.knn2_hamming(q, t)
        assert np.array_equal(vp, vm) and np.array_equal(vm, ov)
        _same_knn(gm, gp)
        _same_knn(gm, o)


def test_knn2_matrix_core_kernel_extreme_rows(vo, oracle):
    """All-zero and all-one rows: distances 0 and 256, the ends of the accumulator's range."""
    q = np.zeros((70, 32), np.uint8)
    q[1::2] = 255
    t = np.zeros((130, 32), np.uint8)
    t[::3] = 255
    t[5, 0] = 1
    g, gv = vo.knn2_hamming(q, t)
    o, ov = oracle.knn2_hamming(q, t)
    assert np.array_equal(gv, ov)
    _same_knn(g, o)
