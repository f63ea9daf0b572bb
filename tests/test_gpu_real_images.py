"""The reference's own test images (tests/golden/kitti/, see its README) through the HIP path, against the
oracle: the flows of src/sfm/test/triangulation_test.cpp (detectFAST -> trackPoints -> updatePose on
000000 / 000001 / 000003 and on the image_0 / image_1 stereo pair) and the descriptor path (ORB ->
matcher()).  Bit-exact keypoints, descriptors, tracked points, matches and masks; E / R / t within 1e-6.
Until this file every GPU parity input was a synthetic planar-world render or noise."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from visual_odometry_amd import api, synth

pytestmark = pytest.mark.gpu

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kitti")
K = synth.KITTI_K  # f = 718.856, c = (607.1928, 185.2157): triangulation_test.cpp:15-16


def load(name):
    im = np.array(Image.open(os.path.join(HERE, name)))
    assert im.shape == (376, 1241) and im.dtype == np.uint8
    return np.ascontiguousarray(im)


@pytest.fixture(scope="module")
def images():
    return {n: load(n + ".png") for n in ("000000", "000001", "000003", "image_1_000000")}


def test_fast_on_kitti(vo, oracle, images):
    for name, im in images.items():
        g, o = vo.fast_detect(im, 20, True), oracle.fast_detect(im, 20, True)
        assert 1500 < len(o) < 8000 and np.array_equal(g, o), name


@pytest.mark.parametrize("nfeatures", [2000, 5000])  # cuda variant / cpu variant of the reference (feature_detector.cpp:5)
def test_orb_on_kitti(vo, oracle, images, nfeatures):
    for name in ("000000", "000003"):
        kg, dg = vo.orb_detect_compute(images[name], api.orb_params(nfeatures))
        ko, do = oracle.orb_detect_compute(images[name], oracle.orb_params(nfeatures))
        assert len(ko) >= nfeatures * 0.9
        assert np.array_equal(kg, ko) and np.array_equal(dg, do), name
    # per level: pyramid and blurred pyramid (real gradients through the resize and blur kernels)
    for level in (0, 3, 7):
        for which, blurred in ((api.VoContext.debug_read_level, False), (api.VoContext.debug_read_level, True)):
            got = vo.debug_read_level(1 if blurred else 0, 0, level)
            want = oracle.orb_level(images["000003"], oracle.orb_params(nfeatures), level, blurred=blurred)
            assert np.array_equal(got, want), (level, blurred)


def _lk_pose(vo, oracle, a, b):
    """detectFAST -> trackPoints -> removePoints -> updatePose, the flow of triangulation_test.cpp:40-66."""
    kp = vo.fast_detect(a, 20, True)
    assert np.array_equal(kp, oracle.fast_detect(a, 20, True))
    p0 = np.stack([kp["x"], kp["y"]], 1).astype(np.float32)
    g1, gs = vo.lk_track(a, b, p0, api.lk_params())
    o1, os_ = oracle.lk_track(a, b, p0, oracle.lk_params())
    assert np.array_equal(gs, os_) and np.array_equal(g1, o1)
    q0, q1, _, _, _, _ = oracle.remove_points(p0, o1, os_)
    assert len(q0) > 500
    rc_g, E_g, m_g = vo.find_essential(q0, q1, K, api.LMEDS)
    rc_o, E_o, m_o, _ = oracle.find_essential(q0, q1, K, oracle.LMEDS)
    assert rc_g == rc_o == 0 and np.array_equal(m_g, m_o)
    assert min(np.abs(E_g - E_o).max(), np.abs(E_g + E_o).max()) < 1e-6
    ng, R, t, pm, _ = vo.recover_pose(E_g, q0, q1, K, 200.0, m_g)
    no, Ro, to, pmo, _ = oracle.recover_pose(E_o, q0, q1, K, 200.0, m_o)
    assert ng == no and np.array_equal(pm, pmo)
    assert np.abs(R - Ro).max() < 1e-6 and np.abs(t - to).max() < 1e-6
    return R, t, ng


def test_default_pipeline_on_consecutive_kitti_frames(vo, oracle, images):
    R, t, ng = _lk_pose(vo, oracle, images["000000"], images["000001"])
    assert ng > 200
    # the car drives forward: x2 = R x1 + t with t along -z, hardly any rotation between 10 Hz frames
    assert t[2] < -0.95 and np.abs(R - np.eye(3)).max() < 0.02
    R, t, ng = _lk_pose(vo, oracle, images["000000"], images["000003"])
    assert ng > 100 and t[2] < -0.9


def test_default_pipeline_on_the_stereo_pair(vo, oracle, images):
    R, t, ng = _lk_pose(vo, oracle, images["000000"], images["image_1_000000"])
    # the right camera sits 0.54 m along +x of the left one: camera-to-world translation +x, i.e. t = -R^T.. ~ (-1, 0, 0)
    assert ng > 100 and t[0] < -0.98 and np.abs(R - np.eye(3)).max() < 0.02


def test_descriptor_path_on_kitti(oracle, images):
    fr = np.ascontiguousarray(np.stack([images["000000"], images["000001"], images["000003"]]))
    ctx = api.VoContext(0)
    try:
        d = torch.from_numpy(fr).to("cuda:0")
        torch.cuda.synchronize()
        res = ctx.pairs_run_dev(d.data_ptr(), 3, 1241, 376, 1241, 1241 * 376, K, api.pair_params(2000))
        po = oracle.orb_params(2000)
        feats = [oracle.orb_detect_compute(f, po) for f in fr]
        for p in range(2):
            (k0, d0), (k1, d1) = feats[p], feats[p + 1]
            kg, dg = ctx.pairs_get_frame(p)
            assert np.array_equal(kg, k0) and np.array_equal(dg, d0)
            knn, nv = oracle.knn2_hamming(d0, d1)
            good = oracle.match_filter(knn, nv, k0, k1, 0.7, 200.0)
            gg, m1, m2 = ctx.pairs_get_matches(p)
            assert len(good) > 150 and np.array_equal(gg, good)
            p0 = np.stack([k0["x"][good["queryIdx"]], k0["y"][good["queryIdx"]]], 1)
            p1 = np.stack([k1["x"][good["trainIdx"]], k1["y"][good["trainIdx"]]], 1)
            rc, E, mask, _ = oracle.find_essential(p0, p1, K, oracle.RANSAC)
            r = res[p]
            assert rc == r.status == 0 and np.array_equal(m1, mask)
            Eg = np.array(r.E).reshape(3, 3)
            assert min(np.abs(Eg - E).max(), np.abs(Eg + E).max()) < 1e-6
            ng, R, t, pm, _ = oracle.recover_pose(E, p0, p1, K, 200.0, mask)
            assert r.n_pose == ng and np.array_equal(m2, pm)
            assert np.abs(np.array(r.R).reshape(3, 3) - R).max() < 1e-6 and np.abs(np.array(r.t) - t).max() < 1e-6
            assert np.array(r.t)[2] < -0.9  # forward motion again
    finally:
        ctx.close()
