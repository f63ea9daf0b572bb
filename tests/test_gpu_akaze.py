"""AKAZE through the HIP path against the oracle's restatement (oracle/oracle_akaze.cpp;
OpenCV itself is absent: parity unpinned, see that file's header).  A float pipeline evaluated in the same
operation order on both sides: every plane of the non-linear scale space, the keypoints and the 486-bit MLDB
descriptors are compared BIT FOR BIT; the only float tolerance is the keypoint angle (fastAtan2 of two sums)."""
import os

import numpy as np
import pytest
from PIL import Image

from visual_odometry_amd import api, synth

pytestmark = pytest.mark.gpu
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kitti")


def load(name):
    return np.ascontiguousarray(np.array(Image.open(os.path.join(HERE, name + ".png"))))


def check_planes(vo, oracle, im, po, levels):
    names = ["Lt", "Lsmooth", "Lx", "Ly", "Ldet", "mask"]
    for lvl in levels:
        for which in range(6):
            g = vo.debug_akaze_level(lvl, which)
            o = oracle.akaze_level(im, po, lvl, which)
            assert g.shape == o.shape
            assert np.array_equal(g.view(np.uint32), o.view(np.uint32)), (
                "level %d %s: %d of %d floats differ, max |diff| %g" % (
                    lvl, names[which], (g != o).sum(), g.size, np.abs(g - o).max()))


def check_features(kg, dg, ko, do):
    assert len(kg) == len(ko) and len(ko) > 0
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        assert np.array_equal(kg[f], ko[f]), f
    assert np.array_equal(kg["angle"], ko["angle"]), "angles"
    assert dg.shape == do.shape == (len(ko), 61)
    assert np.array_equal(dg, do), "%d descriptor rows differ" % (dg != do).any(1).sum()
    assert not (dg[:, 60] & 0xC0).any()  # 486 bits: the top two of byte 60 stay clear


def test_akaze_scale_space_planes_small(vo, oracle):
    """Every plane of every level on a small synthetic frame (the oracle rebuilds its scale space per call)."""
    K = synth.KITTI_K.copy()
    K[0, 2], K[1, 2] = 320.0, 96.0
    frames, _, _ = synth.make_sequence(5, 1, w=640, h=192, K=K)
    im = frames[0]
    kg, dg = vo.akaze_detect_compute(im)
    po = oracle.akaze_params()
    lay = oracle.akaze_layout(640, 192, po)
    assert len(lay["w"]) == 12  # 640 x 192: three octaves of four sublevels (80 x 24 is below cv's 80 x 40 floor)
    check_planes(vo, oracle, im, po, range(len(lay["w"])))
    ko, do = oracle.akaze_detect_compute(im, po)
    check_features(kg, dg, ko, do)


@pytest.mark.parametrize("name", ["000000", "000003"])
def test_akaze_on_kitti(vo, oracle, name):
    im = load(name)
    kg, dg = vo.akaze_detect_compute(im)
    po = oracle.akaze_params()
    ko, do = oracle.akaze_detect_compute(im, po)
    assert 800 < len(ko) < 6000
    check_features(kg, dg, ko, do)
    if name == "000000":
        check_planes(vo, oracle, im, po, [0, 3, 4, 9, 15])


def test_akaze_parameters_and_odd_sizes(vo, oracle):
    frames, _, _ = synth.make_sequence(9, 1, w=517, h=203, K=synth.KITTI_K)
    im = frames[0]
    for kw in (dict(threshold=0.0003), dict(n_octaves=2, n_sublevels=3), dict(channels=1), dict(channels=2),
               dict(pattern_size=8)):
        kg, dg = vo.akaze_detect_compute(im, api.akaze_params(**kw))
        ko, do = oracle.akaze_detect_compute(im, oracle.akaze_params(**kw))
        check_features(kg, dg, ko, do)
    # a constant image has no keypoints (and exercises the hmax == 0 branch of the contrast estimate)
    kg, dg = vo.akaze_detect_compute(np.full((203, 517), 90, np.uint8))
    assert len(kg) == 0
    # capacity: the true count comes back with VO_ERR_CAPACITY
    with pytest.raises(RuntimeError):
        vo.akaze_detect_compute(im, cap=5)
    with pytest.raises(RuntimeError):
        vo.akaze_detect_compute(im, api.akaze_params(pattern_size=7))


def test_akaze_graph_replay_equals_plain_launches(vo):
    """Phase 1 runs as a captured hipGraph by default; VO_OPT_AKAZE_GRAPH = 0 queues the same kernels one by one.
    Same keypoints and descriptors, also when the graph is re-captured for another size and replayed again."""
    a, b = load("000000"), np.ascontiguousarray(load("000003")[:300, :900])
    ref = {}
    vo.set_option(api.OPT_AKAZE_GRAPH, 0)
    for name, im in (("a", a), ("b", b)):
        ref[name] = vo.akaze_detect_compute(im)
    vo.set_option(api.OPT_AKAZE_GRAPH, 1)
    for name, im in (("a", a), ("b", b), ("a", a), ("a", a)):
        k, d = vo.akaze_detect_compute(im)
        assert np.array_equal(k, ref[name][0]) and np.array_equal(d, ref[name][1]), name


def test_akaze_descriptors_match_across_frames(vo):
    """61-byte rows straight into the matcher (padded to 64 inside): consecutive KITTI frames agree on most
    of their keypoints at a small Hamming distance."""
    k0, d0 = vo.akaze_detect_compute(load("000000"))
    k1, d1 = vo.akaze_detect_compute(load("000001"))
    knn, nv = vo.knn2_hamming(d0, d1)
    best = knn[:, 0]["distance"]
    ratio_ok = best < 0.7 * knn[:, 1]["distance"]
    assert ratio_ok.mean() > 0.3 and np.median(best[ratio_ok]) < 60
    p0 = np.stack([k0["x"], k0["y"]], 1)[ratio_ok]
    p1 = np.stack([k1["x"], k1["y"]], 1)[knn[:, 0]["trainIdx"][ratio_ok]]
    assert np.median(np.linalg.norm(p0 - p1, axis=1)) < 30  # forward motion: a few pixels of flow


def test_akaze_strided_input_and_bad_arguments(vo):
    """Rows with padding (stride > width) give the result of the tight image; bad arguments are refused with
    VO_ERR_INVALID and leave the context usable."""
    import ctypes as C

    from visual_odometry_amd.api import KP_DTYPE, _p

    im = load("000001")[:200, :640]
    tight = np.ascontiguousarray(im)
    padded = np.zeros((200, 704), np.uint8)
    padded[:, :640] = tight
    k0, d0 = vo.akaze_detect_compute(tight)
    prm = api.akaze_params()
    kp = np.zeros(8192, KP_DTYPE)
    desc = np.zeros((8192, 61), np.uint8)
    n = C.c_int(0)
    rc = vo.L.vo_akaze_detect_compute(vo.h, _p(padded), 640, 200, 704, C.byref(prm), _p(kp), _p(desc), 8192, C.byref(n))
    assert rc == 0 and n.value == len(k0)
    assert np.array_equal(kp[: n.value], k0) and np.array_equal(desc[: n.value], d0)
    for args in ((None, 640, 200, 704), (_p(padded), 8, 200, 704), (_p(padded), 640, 200, 600)):
        rc = vo.L.vo_akaze_detect_compute(vo.h, args[0], args[1], args[2], args[3], C.byref(prm), _p(kp), _p(desc),
                                          8192, C.byref(n))
        assert rc == 1, rc  # VO_ERR_INVALID
    k1, d1 = vo.akaze_detect_compute(tight)
    assert np.array_equal(k1, k0) and np.array_equal(d1, d0)
