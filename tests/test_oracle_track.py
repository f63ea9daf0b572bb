"""CPU tests of the tracker restatement (oracle/oracle_track.cpp): hand-derived known answers for
cv::pyrDown and the Scharr derivatives, level sizes of buildOpticalFlowPyramid, sub-pixel recovery
of known translations, and removePoints() (reference: src/features/utils.h:9-58).
No OpenCV in this image: parity with cv itself is unpinned (oracle/vo_oracle.h)."""
import numpy as np

from visual_odometry_amd import synth


def test_lk_level_sizes(oracle):
    p = oracle.lk_params()
    assert oracle.lk_levels(1241, 376, p) == [(1241, 376), (621, 188), (311, 94), (156, 47)]
    # a level is not built when it would not be larger than the window (buildOpticalFlowPyramid)
    assert oracle.lk_levels(80, 60, p) == [(80, 60), (40, 30)]
    assert oracle.lk_levels(40, 40, p) == [(40, 40)]
    assert oracle.lk_levels(1241, 376, oracle.lk_params(max_level=1)) == [(1241, 376), (621, 188)]


def test_pyrdown_known_answers(oracle):
    p = oracle.lk_params()
    # constant image stays constant: (256 v + 128) >> 8 = v
    img = np.full((48, 64), 77, np.uint8)
    l1, _ = oracle.lk_level(img, p, 1)
    assert l1.shape == (24, 32) and (l1 == 77).all()
    # horizontal ramp v = 2x: the binomial kernel is symmetric, so interior pixels keep the centre value
    img = np.tile((2 * np.arange(64)).astype(np.uint8), (48, 1))
    l1, _ = oracle.lk_level(img, p, 1)
    assert (l1[:, 1:31] == (4 * np.arange(1, 31))[None, :]).all()
    # left edge, reflect-101: taps at x = -2..2 read 4,2,0,2,4 -> (4 + 4*2 + 0 + 4*2 + 4)/16 = 1.5 -> 2 after rounding
    assert (l1[:, 0] == 2).all()
    # impulse 255 at an even position: centre weight 36/256 -> (255*36 + 128) >> 8 = 36
    img = np.zeros((48, 64), np.uint8)
    img[20, 30] = 255
    l1, _ = oracle.lk_level(img, p, 1)
    assert l1[10, 15] == 36 and l1[10, 14] == (255 * 6 + 128) >> 8 and l1[9, 14] == (255 * 1 + 128) >> 8


def test_scharr_known_answers(oracle):
    p = oracle.lk_params()
    # ramp v = 3x + 2y: dx = 2*3*(3+10+3) = 96, dy = 2*2*16 = 64 in the interior
    y, x = np.mgrid[0:40, 0:50]
    img = (3 * x + 2 * y).astype(np.uint8)
    _, d = oracle.lk_level(img, p, 0)
    assert (d[1:-1, 1:-1, 0] == 96).all() and (d[1:-1, 1:-1, 1] == 64).all()
    # reflect-101 at the edges: the two neighbours coincide, the difference vanishes
    assert (d[1:-1, 0, 0] == 0).all() and (d[1:-1, -1, 0] == 0).all()
    assert (d[0, 1:-1, 1] == 0).all() and (d[-1, 1:-1, 1] == 0).all()


def _shift_pair(dx, dy, seed=3, w=320, h=240):
    """A smooth random texture sampled at two sub-pixel offsets (bilinear, float), as 8-bit images."""
    rng = np.random.default_rng(seed)
    big = rng.random((h // 8 + 6, w // 8 + 6)).astype(np.float64)
    def sample(ox, oy):
        ys = (np.arange(h) + oy) / 8.0 + 2
        xs = (np.arange(w) + ox) / 8.0 + 2
        y0 = np.floor(ys).astype(int); x0 = np.floor(xs).astype(int)
        fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
        # cubic-ish smoothing via two bilinear passes keeps gradients well defined
        a = big[y0][:, x0] * (1 - fy) * (1 - fx) + big[y0][:, x0 + 1] * (1 - fy) * fx
        b = big[y0 + 1][:, x0] * fy * (1 - fx) + big[y0 + 1][:, x0 + 1] * fy * fx
        return np.clip((a + b) * 255, 0, 255).astype(np.uint8)
    return sample(0, 0), sample(-dx, -dy)


def test_lk_recovers_translation(oracle):
    for dx, dy in [(0.0, 0.0), (1.5, -0.75), (6.25, 3.5), (-11.0, 2.0)]:
        a, b = _shift_pair(dx, dy)
        ys, xs = np.mgrid[40:200:20, 40:280:20]
        p0 = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.float32)
        for ch in (1, 3):
            p1, st = oracle.lk_track(a, b, p0, oracle.lk_params(channels=ch))
            assert st.all()
            err = np.abs(p1 - p0 - np.array([dx, dy], np.float32)).max()
            assert err < 0.15, (dx, dy, ch, err)


def test_lk_status_and_borders(oracle):
    a, b = _shift_pair(2.0, 1.0)
    h, w = a.shape
    p0 = np.array([[-40.0, 10.0], [w + 30.0, 50.0], [5.0, 5.0], [w - 3.0, h - 3.0], [160.0, 120.0]], np.float32)
    p1, st = oracle.lk_track(a, b, p0, oracle.lk_params())
    assert st[0] == 0 and st[1] == 0  # window entirely outside the image at level 0
    assert st[4] == 1
    # a flat image has no texture: minEig below the threshold at every level, status 0, point unmoved
    flat = np.full((120, 160), 90, np.uint8)
    p1, st = oracle.lk_track(flat, flat, np.array([[80.0, 60.0]], np.float32), oracle.lk_params())
    assert st[0] == 0 and np.array_equal(p1[0], [80.0, 60.0])


def test_lk_three_channels_lowers_the_eigenvalue_bar(oracle):
    # weak texture: amplitude so small that the single-channel minimum eigenvalue is below 1e-4
    # at level 0 while three identical channels (x3 sums) pass - the two modes differ as cv's do
    rng = np.random.default_rng(0)
    img = (100 + 3 * (rng.random((120, 160)) < 0.05)).astype(np.uint8)
    pts = np.stack(np.meshgrid(np.arange(30, 130, 10), np.arange(30, 90, 10)), -1).reshape(-1, 2).astype(np.float32)
    _, s1 = oracle.lk_track(img, img, pts, oracle.lk_params(channels=1))
    _, s3 = oracle.lk_track(img, img, pts, oracle.lk_params(channels=3))
    assert (s3 >= s1).all() and s3.sum() > s1.sum()


def test_remove_points_matches_the_reference_loop(oracle):
    """Literal restatement of the erase-from-the-back loop in python against the compaction."""
    rng = np.random.default_rng(11)
    n = 200
    p0 = (rng.random((n, 2)) * 300).astype(np.float32)
    p1 = p0 + rng.normal(0, 2, (n, 2)).astype(np.float32)
    p1[5] = [-1.0, 20.0]
    p1[17] += 80.0  # an outlier displacement
    status = (rng.random(n) > 0.1).astype(np.uint8)
    pts3d = rng.random((150, 3))
    mask = rng.integers(0, 2, 170).astype(np.uint8)

    # reference semantics, element by element (src/features/utils.h:11-57)
    d = np.array([np.float32(np.sqrt(float(a[0] - b[0]) ** 2 + float(a[1] - b[1]) ** 2)) for a, b in zip(p0, p1)], np.float32)
    dists = sorted(d.tolist())
    mean = sum(dists) / len(dists)
    stdev = np.sqrt(sum(x * x for x in dists) / len(dists) - mean * mean)
    L0, L1, L3, LM, ST = p0.tolist(), p1.tolist(), pts3d.tolist(), mask.tolist(), status.copy()
    for i in range(n - 1, -1, -1):
        x, y = L1[i]
        if ST[i] == 0 or x < 0 or y < 0 or abs(float(d[i]) - mean) > stdev * 2:
            if x < 0 or y < 0:
                ST[i] = 0
            del L0[i]
            del dists[i]
            if i < len(L3):
                del L3[i]
            if i < len(LM):
                del LM[i]
            del L1[i]
    ret = int(dists[len(dists) // 2])

    q0, q1, st, q3, qm, r = oracle.remove_points(p0, p1, status, pts3d, mask)
    assert np.array_equal(q0, np.array(L0, np.float32)) and np.array_equal(q1, np.array(L1, np.float32))
    assert np.array_equal(st, ST)
    assert np.array_equal(q3, np.array(L3)) and np.array_equal(qm, np.array(LM, np.uint8))
    assert r == ret
