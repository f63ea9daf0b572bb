"""GPU parity: FAST-9/16 and ORB through the C ABI against the CPU oracle — bit-exact keypoint
coordinates, responses, angles, order, and descriptor bits.
Mirrors the reference's FeatureDetector calls (src/features/feature_detector.cpp:12-42)."""
import numpy as np
import pytest

from visual_odometry_amd import synth

pytestmark = pytest.mark.gpu


def _noise(h, w, seed, smooth=False):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    if smooth:
        f = img.astype(np.float32)
        f = (f + np.roll(f, 1, 0) + np.roll(f, 1, 1) + np.roll(f, (1, 1), (0, 1))) / 4
        img = f.astype(np.uint8)
    return img


def test_fast_kitti_frame(vo, oracle, frames):
    fr = frames[0]
    for k in range(2):
        g = vo.fast_detect(fr[k], 20, True)
        o = oracle.fast_detect(fr[k], 20, True)
        assert len(o) > 1000
        assert np.array_equal(g, o)


@pytest.mark.parametrize("h,w", [(7, 7), (8, 9), (16, 64), (33, 65), (100, 129), (376, 1241)])
@pytest.mark.parametrize("thr", [0, 20, 100])
def test_fast_noise_sizes(vo, oracle, h, w, thr):
    img = _noise(h, w, h * 1000 + w + thr, smooth=(thr == 0))
    g = vo.fast_detect(img, thr, True, cap=h * w)
    o = oracle.fast_detect(img, thr, True)
    assert np.array_equal(g, o)


@pytest.mark.parametrize("thr", [200, 254, 255, 300])
def test_fast_threshold_extremes(vo, oracle, thr):
    """Black / white blocks with isolated pixels: differences of exactly 255 exist, so thresholds up to 254 still
    find corners, 255 finds none, and a threshold beyond the byte range is clamped to 255 as cv does.  The
    kernel's packed two-field test (r - v + t | v + t - r << 16) has its largest fields here."""
    rng = np.random.default_rng(thr)
    img = (rng.integers(0, 2, (90, 160)) * 255).astype(np.uint8)
    img = np.kron(img[::3, ::4], np.ones((3, 4), np.uint8))[:90, :160].copy()
    img[rng.integers(0, 90, 300), rng.integers(0, 160, 300)] ^= 255
    g = vo.fast_detect(img, thr, True, cap=90 * 160)
    o = oracle.fast_detect(img, min(thr, 255), True)
    assert np.array_equal(g, o)
    assert (len(o) > 0) == (thr < 255)


def test_fast_without_nms(vo, oracle):
    img = _noise(120, 200, 9, smooth=True)
    g = vo.fast_detect(img, 20, False, cap=120 * 200)
    o = oracle.fast_detect(img, 20, False)
    assert len(o) > 0 and np.array_equal(g, o)


def test_fast_too_small_and_flat(vo, oracle):
    assert len(vo.fast_detect(np.zeros((6, 40), np.uint8))) == 0
    assert len(vo.fast_detect(np.full((50, 60), 128, np.uint8))) == 0


def test_fast_strided_input(vo, oracle):
    big = _noise(90, 300, 4, smooth=True)
    view = big[:, 17:217]  # non-contiguous rows: stride 300, width 200
    g = vo.fast_detect(np.ascontiguousarray(view), 20, True)
    o = oracle.fast_detect(np.ascontiguousarray(view), 20, True)
    assert np.array_equal(g, o)


def test_fast_capacity_error(vo, frames):
    from visual_odometry_amd.api import VoError

    with pytest.raises(VoError) as e:
        vo.fast_detect(frames[0][0], 20, True, cap=10)
    assert e.value.code == 2


@pytest.mark.parametrize("nfeatures", [2000, 5000, 500])
def test_orb_detect_compute_kitti_frame(vo, oracle, frames, nfeatures):
    # 5000 = the reference's CPU FeatureDetector, 2000 = its CUDA variant / BASELINE config
    from visual_odometry_amd import api

    fr = frames[0]
    pg = api.orb_params(nfeatures)
    po = oracle.orb_params(nfeatures)
    kg, dg = vo.orb_detect_compute(fr[1], pg)
    ko, do = oracle.orb_detect_compute(fr[1], po)
    assert len(ko) >= nfeatures * 0.9
    assert len(kg) == len(ko)
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(kg[f], ko[f]), f
    assert np.array_equal(dg, do)


def test_orb_small_image_few_levels(vo, oracle):
    from visual_odometry_amd import api

    img = _noise(200, 320, 77, smooth=True)
    kg, dg = vo.orb_detect_compute(img, api.orb_params(300, nlevels=3))
    ko, do = oracle.orb_detect_compute(img, oracle.orb_params(300, nlevels=3))
    assert len(ko) > 50
    assert np.array_equal(kg, ko) and np.array_equal(dg, do)


@pytest.mark.parametrize("h,w,edge", [(376, 1241, 19), (376, 1241, 45), (376, 1241, 64), (150, 500, 31), (70, 300, 31)])
def test_orb_edge_thresholds_and_empty_levels(vo, oracle, frames, h, w, edge):
    # the detector only visits the region the border filter keeps: other borders move that region, and
    # levels not larger than twice the border (the top ones here) have no region at all
    from visual_odometry_amd import api

    img = np.ascontiguousarray(frames[0][1][:h, :w])
    kg, dg = vo.orb_detect_compute(img, api.orb_params(1000, edge_threshold=edge))
    ko, do = oracle.orb_detect_compute(img, oracle.orb_params(1000, edge_threshold=edge))
    assert len(ko) > (20 if h > 100 else 1)
    assert np.array_equal(kg, ko) and np.array_equal(dg, do)
    # and back to the default border on the same context (the geometry is rebuilt)
    kg, dg = vo.orb_detect_compute(img, api.orb_params(1000))
    ko, do = oracle.orb_detect_compute(img, oracle.orb_params(1000))
    assert np.array_equal(kg, ko) and np.array_equal(dg, do)


def test_orb_full_hd_frame(oracle, frames):
    # a frame four times KITTI's area on a context with wider limits: the mask table no longer fits the
    # row scan's register-resident path, the resize tiles and selection lists are larger
    from visual_odometry_amd import api

    img = np.ascontiguousarray(np.tile(frames[0][1], (3, 2))[:1080, :1920])
    ctx = api.VoContext(0, max_width=1920, max_height=1080)
    try:
        kg, dg = ctx.orb_detect_compute(img, api.orb_params(2000, fast_threshold=60))
        kf = ctx.fast_detect(img, 60, True, cap=200000)
    finally:
        ctx.close()
    ko, do = oracle.orb_detect_compute(img, oracle.orb_params(2000, fast_threshold=60))
    assert len(ko) >= 1800
    assert np.array_equal(kg, ko) and np.array_equal(dg, do)
    fo = oracle.fast_detect(img, 60, True)
    assert len(fo) > 5000 and np.array_equal(kf, fo)


def test_orb_more_corners_than_the_candidate_list(oracle):
    # a noise image has more FAST corners per frame (~78 k) than the initial candidate list (65 536): the
    # list grows and the frame goes through the host selection; cv has no such limit
    from visual_odometry_amd import api

    img = _noise(376, 1241, 24, smooth=True)
    ctx = api.VoContext(0)
    try:
        kg, dg = ctx.orb_detect_compute(img, api.orb_params(2000, edge_threshold=19))
        assert ctx.get_option(api.OPT_LAST_FLAGS) & 1  # the device selection handed the frame over
        assert ctx.get_option(api.OPT_CAND_CAP) > 65536  # the lists grew
        kf = ctx.fast_detect(img, 20, True, cap=400000)
        # and a frame with few corners afterwards on the same (grown) context
        img2 = _noise(200, 320, 77, smooth=True)
        kg2, dg2 = ctx.orb_detect_compute(img2, api.orb_params(300, nlevels=3))
    finally:
        ctx.close()
    ko, do = oracle.orb_detect_compute(img, oracle.orb_params(2000, edge_threshold=19))
    assert np.array_equal(kg, ko) and np.array_equal(dg, do)
    fo = oracle.fast_detect(img, 20, True)
    assert np.array_equal(kf, fo)
    ko2, do2 = oracle.orb_detect_compute(img2, oracle.orb_params(300, nlevels=3))
    assert np.array_equal(kg2, ko2) and np.array_equal(dg2, do2)


def test_orb_flat_image_gives_nothing(vo, oracle):
    from visual_odometry_amd import api

    kg, dg = vo.orb_detect_compute(np.full((376, 1241), 90, np.uint8), api.orb_params(2000))
    assert len(kg) == 0 and dg.shape == (0, 32)


def test_orb_compute_on_given_points(vo, oracle, frames):
    # computeORB(): default cv::KeyPoint (angle -1, octave 0) at tracked points
    # (reference: src/features/feature_detector.cpp:32-42)
    from visual_odometry_amd import api

    fr = frames[0]
    rng = np.random.default_rng(11)
    n = 1500
    kps = np.zeros(n, oracle.KP_DTYPE)
    kps["x"] = rng.uniform(-5, 1246, n).astype(np.float32)
    kps["y"] = rng.uniform(-5, 381, n).astype(np.float32)
    kps["angle"] = -1
    kps["class_id"] = -1
    kg, dg = vo.orb_compute(fr[0], api.orb_params(5000), kps)
    ko, do = oracle.orb_compute(fr[0], oracle.orb_params(5000), kps)
    assert 0 < len(ko) < n  # the border filter dropped some
    assert np.array_equal(kg, ko) and np.array_equal(dg, do)


def test_orb_compute_with_octaves_and_angles(vo, oracle, frames):
    from visual_odometry_amd import api

    fr = frames[0]
    ko0, _ = oracle.orb_detect_compute(fr[2], oracle.orb_params(800))
    kg, dg = vo.orb_compute(fr[2], api.orb_params(800), ko0)
    ko, do = oracle.orb_compute(fr[2], oracle.orb_params(800), ko0)
    assert np.array_equal(kg, ko) and np.array_equal(dg, do)


def test_orb_stress_8k(vo, oracle, frames):
    from visual_odometry_amd import api

    fr = frames[0]
    kg, dg = vo.orb_detect_compute(fr[0], api.orb_params(8192), cap=20000)
    ko, do = oracle.orb_detect_compute(fr[0], oracle.orb_params(8192))
    assert np.array_equal(kg, ko) and np.array_equal(dg, do)


def test_orb_device_selection_is_used_and_host_path_agrees(oracle, frames):
    """The keypoint selection (retainBest = nth_element + partition, whose permutation fixes the
    keypoint order) runs on the device; a context with a tiny selection buffer is forced through
    the host path.  Both must give the oracle's keypoints in the oracle's order."""
    from visual_odometry_amd import api

    fr = frames[0]
    ko, do = oracle.orb_detect_compute(fr[0], oracle.orb_params(2000))
    ctx = api.VoContext(0)
    try:
        kg, dg = ctx.orb_detect_compute(fr[0], api.orb_params(2000))
        assert ctx.get_option(api.OPT_LAST_FLAGS) == 0
        assert np.array_equal(kg, ko) and np.array_equal(dg, do)
        ctx.set_option(api.OPT_SEL_CAP, 64)
        kh, dh = ctx.orb_detect_compute(fr[0], api.orb_params(2000))
        assert ctx.get_option(api.OPT_LAST_FLAGS) & 1
        assert np.array_equal(kh, ko) and np.array_equal(dh, do)
    finally:
        ctx.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_orb_periodic_image_response_ties(oracle, seed):
    """A periodic texture gives many keypoints with identical FAST scores and Harris responses: the
    ties exercise the equal-key behaviour of the Hoare partition and of partition(>= ambiguous) in
    the device-side selection (asserted to be the path taken)."""
    from visual_odometry_amd import api

    rng = np.random.default_rng(seed)
    cell = rng.integers(0, 256, (94, 106)).astype(np.float32)
    for _ in range(16):
        cell = (cell + np.roll(cell, 1, 0) + np.roll(cell, -1, 0) + np.roll(cell, 1, 1) + np.roll(cell, -1, 1)) / 5
    cell = np.clip((cell - cell.mean()) * 4.0 + 128, 0, 255).astype(np.uint8)
    img = np.tile(cell, (4, 12))[:376, :1241].copy()
    img[100:140, 300:360] = rng.integers(0, 256, (40, 60), dtype=np.uint8)  # break the symmetry a little
    ctx = api.VoContext(0)
    try:
        ctx.set_option(api.OPT_SEL_CAP, 13000)
        for nf in (500, 2000):
            kg, dg = ctx.orb_detect_compute(img, api.orb_params(nf), cap=40000)
            ko, do = oracle.orb_detect_compute(img, oracle.orb_params(nf))
            if nf == 2000:  # tied responses at the cut: more than nfeatures keypoints are kept
                assert len(ko) > nf and len(np.unique(ko["response"])) < len(ko)
            assert ctx.get_option(api.OPT_LAST_FLAGS) == 0
            assert np.array_equal(kg, ko) and np.array_equal(dg, do)
    finally:
        ctx.close()


def test_orb_pyramid_and_blur_levels(vo, oracle, frames):
    """Intermediate images of the ORB path, level by level: the INTER_LINEAR_EXACT resize chain and
    the 7x7 fixed-point Gaussian blur, bit-exact against the oracle."""
    from visual_odometry_amd import api

    fr = frames[0]
    vo.orb_detect_compute(fr[2], api.orb_params(2000))
    po = oracle.orb_params(2000)
    for level in range(8):
        for which, blurred in ((0, False), (1, True)):
            g = vo.debug_read_level(which, 0, level)
            o = oracle.orb_level(fr[2], po, level, blurred=blurred)
            assert g.shape == o.shape
            assert np.array_equal(g, o), "level %d %s: %d pixels differ" % (
                level, "blur" if blurred else "pyramid", int((g != o).sum()))


@pytest.mark.parametrize("scale,nlevels,w", [(2.0, 4, 1241), (1.5, 5, 1241), (1.1, 8, 1241), (1.2, 8, 1024), (1.2, 6, 337)])
def test_orb_other_scale_factors_and_widths(vo, oracle, frames, scale, nlevels, w):
    """Scale factors other than 1.2 size the resize kernel's LDS tile differently (2.0 needs more than the
    default 64 KB of dynamic LDS), widths that are or are not multiples of 16 move the level-0 copy's chunk
    boundaries: every pyramid level bit-exact, then keypoints and descriptors."""
    from visual_odometry_amd import api

    img = np.ascontiguousarray(frames[0][2][:, :w])
    pg = api.orb_params(1500, scale_factor=scale, nlevels=nlevels)
    po = oracle.orb_params(1500, scale_factor=scale, nlevels=nlevels)
    kg, dg = vo.orb_detect_compute(img, pg)
    for level in range(nlevels):
        g = vo.debug_read_level(0, 0, level)
        o = oracle.orb_level(img, po, level, blurred=False)
        assert g.shape == o.shape and np.array_equal(g, o), "level %d" % level
    ko, do = oracle.orb_detect_compute(img, po)
    assert len(ko) > 300
    assert np.array_equal(kg, ko) and np.array_equal(dg, do)


def test_orb_edge_threshold_below_the_window_radius_is_refused(vo, frames):
    """Keypoints closer than 19 pixels to the border would have their descriptor window outside the image."""
    from visual_odometry_amd import api
    from visual_odometry_amd.api import VoError

    for edge in (0, 8, 18):
        with pytest.raises(VoError) as e:
            vo.orb_detect_compute(frames[0][1], api.orb_params(500, edge_threshold=edge))
        assert e.value.code == 1
    assert len(vo.orb_detect_compute(frames[0][1], api.orb_params(500, edge_threshold=19))[0]) > 400


@pytest.mark.parametrize("edge", [31, 19])
def test_orb_fused_orientation_description_equals_the_two_kernels(vo, oracle, frames, edge):
    """The detection path computes the IC angle and the rBRIEF descriptor in one kernel (k_orient_rbrief);
    VO_OPT_ORB_SPLIT = 1 runs k_ic_angle and k_rbrief apart.  Same keypoints (angle included), same descriptors, both
    equal to the oracle, at the default border and at the smallest one accepted."""
    from visual_odometry_amd import api

    img = frames[0][1]
    pg = api.orb_params(2000, edge_threshold=edge)
    vo.set_option(api.OPT_ORB_SPLIT, 1)
    try:
        ks, ds = vo.orb_detect_compute(img, pg)
    finally:
        vo.set_option(api.OPT_ORB_SPLIT, 0)
    kf, df = vo.orb_detect_compute(img, pg)
    assert len(kf) > 1500
    assert np.array_equal(kf, ks) and np.array_equal(df, ds)
    ko, do = oracle.orb_detect_compute(img, oracle.orb_params(2000, edge_threshold=edge))
    assert np.array_equal(kf, ko) and np.array_equal(df, do)


def test_orb_large_sparse_image_list_build_from_long_mask_tables(oracle):
    """A 2600 x 1300 image with a few thousand isolated corners: the detector's mask table of a level has up to
    ~50 000 words, far more than the 8 192 a selection block keeps in registers, so the block builds its list by the
    two-pass loop (count, then read again and scatter) on levels 0-3 and from registers on the small levels; the
    candidate counts stay below the selection buffer, so the device path is the one that runs (asserted)."""
    from visual_odometry_amd import api

    rng = np.random.default_rng(77)
    h, w = 1300, 2600
    yy, xx = np.mgrid[0:h, 0:w]
    img = (60 + 40 * np.sin(xx / 300.0) + 30 * np.cos(yy / 170.0)).astype(np.float32)
    for _ in range(900):
        y, x = int(rng.integers(40, h - 80)), int(rng.integers(40, w - 80))
        hh, ww = int(rng.integers(6, 40)), int(rng.integers(6, 40))
        img[y:y + hh, x:x + ww] += float(rng.integers(50, 120)) * (1 if rng.random() < 0.5 else -1)
    img = np.clip(img, 0, 255).astype(np.uint8)
    ko, do = oracle.orb_detect_compute(img, oracle.orb_params(2000))
    assert len(ko) >= 1500
    ctx = api.VoContext(0, max_width=w, max_height=h)
    try:
        ctx.set_option(api.OPT_SEL_CAP, 13000)
        kg, dg = ctx.orb_detect_compute(img, api.orb_params(2000), cap=8000)
        assert ctx.get_option(api.OPT_LAST_FLAGS) == 0
        assert np.array_equal(kg, ko) and np.array_equal(dg, do)
    finally:
        ctx.close()
