"""Seeded synthetic inputs shaped like the reference's data.

No KITTI data is used or copied: frames are rendered from a piecewise-planar textured world
(ground plane, two side walls, a sky dome) through the KITTI intrinsics the reference's tests
use (reference: src/sfm/test/triangulation_test.cpp:15-16), at 1241x376 8-bit gray, and can be
written out in the KITTI odometry layout the reference's main() reads (reference:
README.md:54-94, src/kitti/kitti.cpp:10-38, src/main.cpp:21-41).
"""
import os

import numpy as np

KITTI_W, KITTI_H = 1241, 376
KITTI_K = np.array([[718.856, 0.0, 607.1928], [0.0, 718.856, 185.2157], [0.0, 0.0, 1.0]])

# sequence lengths of BASELINE.md config 5 (00-10 from the reference's src/kitti/readme.txt:125-135)
KITTI_SEQ_LENGTHS = [4541, 1101, 4661, 801, 271, 2761, 1101, 1101, 4071, 1591, 1201,
                     921, 1061, 3281, 631, 1901, 1731, 491, 1801, 4981, 831, 2721]


def make_texture(seed, size=2048):
    """Corner-rich tileable texture: multi-octave value noise + random axis-aligned rectangles."""
    rng = np.random.default_rng(seed)
    tex = np.zeros((size, size), np.float32)
    for cells, amp in ((8, 40.0), (32, 30.0), (128, 20.0)):
        g = rng.uniform(-1, 1, (cells, cells)).astype(np.float32)
        rep = size // cells
        up = np.kron(g, np.ones((rep, rep), np.float32))
        # cheap smoothing keeps large-scale structure without killing the rectangle corners below
        up = (up + np.roll(up, rep // 2, 0) + np.roll(up, rep // 2, 1) + np.roll(up, (rep // 2, rep // 2), (0, 1))) / 4
        tex += amp * up
    tex += 110.0
    n_rect = size * size // 900
    xs = rng.integers(0, size, n_rect)
    ys = rng.integers(0, size, n_rect)
    ws = rng.integers(6, 40, n_rect)
    hs = rng.integers(6, 40, n_rect)
    vals = rng.integers(10, 245, n_rect)
    for x, y, w, h, v in zip(xs, ys, ws, hs, vals):
        tex[y:min(y + h, size), x:min(x + w, size)] = v
    return np.clip(tex, 0, 255).astype(np.float32)


def _sample(tex, u, v):
    """Bilinear, wrap-around sampling of tex at float coords (u = column, v = row)."""
    size = tex.shape[0]
    u0 = np.floor(u)
    v0 = np.floor(v)
    fu = (u - u0).astype(np.float32)
    fv = (v - v0).astype(np.float32)
    iu = u0.astype(np.int64) % size
    iv = v0.astype(np.int64) % size
    iu1 = (iu + 1) % size
    iv1 = (iv + 1) % size
    a = tex[iv, iu] * (1 - fu) + tex[iv, iu1] * fu
    b = tex[iv1, iu] * (1 - fu) + tex[iv1, iu1] * fu
    return a * (1 - fv) + b * fv


def trajectory(seq_id, n_frames, step=1.0):
    """Camera-to-world poses (n,3,4): ~step m/frame forward (+Z) with a smooth yaw."""
    rng = np.random.default_rng(0x5EED0000 + seq_id)
    f1, f2 = rng.uniform(0.01, 0.03, 2)
    ph1, ph2 = rng.uniform(0, 2 * np.pi, 2)
    poses = np.zeros((n_frames, 3, 4))
    pos = np.zeros(3)
    for k in range(n_frames):
        yaw = 0.08 * np.sin(2 * np.pi * f1 * k + ph1) + 0.04 * np.sin(2 * np.pi * f2 * k + ph2)
        pitch = 0.004 * np.sin(0.11 * k + ph2)
        cy, sy = np.cos(yaw), np.sin(yaw)
        cp, sp = np.cos(pitch), np.sin(pitch)
        Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
        R = Ry @ Rx
        poses[k, :, :3] = R
        poses[k, :, 3] = pos
        pos = pos + R @ np.array([0.0, 0.0, step])
        pos[0] = np.clip(pos[0], -2.5, 2.5)
    return poses


class World:
    """Ground plane Y=1.65, walls X=+-6, sky dome; one texture per surface."""

    def __init__(self, seq_id, tex_size=2048):
        base = 0x5EED0000 + seq_id
        self.ground = make_texture(base + 1, tex_size)
        self.wall_l = make_texture(base + 2, tex_size)
        self.wall_r = make_texture(base + 3, tex_size)
        self.sky = make_texture(base + 4, tex_size)
        self.ppm = 48.0  # texture pixels per metre

    def render(self, pose, K=KITTI_K, w=KITTI_W, h=KITTI_H):
        R = pose[:, :3]
        Cw = pose[:, 3]
        u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
        rc = np.stack([(u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], np.ones_like(u)], -1)
        rw = rc @ R.T
        big = 1e9
        with np.errstate(divide="ignore", invalid="ignore"):
            tg = np.where(rw[..., 1] > 1e-9, (1.65 - Cw[1]) / rw[..., 1], big)
            tl = np.where(rw[..., 0] < -1e-9, (-6.0 - Cw[0]) / rw[..., 0], big)
            tr = np.where(rw[..., 0] > 1e-9, (6.0 - Cw[0]) / rw[..., 0], big)
        tmax = 400.0
        t = np.minimum(np.minimum(tg, tl), tr)
        which = np.where(t >= tmax, 3, np.where(t == tg, 0, np.where(t == tl, 1, 2)))
        P = Cw[None, None, :] + rw * np.minimum(t, tmax)[..., None]
        img = np.zeros((h, w), np.float32)
        ppm = self.ppm
        m = which == 0
        img[m] = _sample(self.ground, P[..., 0][m] * ppm, P[..., 2][m] * ppm)
        m = which == 1
        img[m] = _sample(self.wall_l, P[..., 2][m] * ppm, P[..., 1][m] * ppm)
        m = which == 2
        img[m] = _sample(self.wall_r, P[..., 2][m] * ppm, P[..., 1][m] * ppm)
        m = which == 3
        d = rw[m]
        az = np.arctan2(d[:, 0], d[:, 2])
        el = d[:, 1] / np.linalg.norm(d, axis=1)
        img[m] = _sample(self.sky, az * 1500.0, el * 1500.0) * 0.5 + 90.0
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)


_render_job = None


def _render_one(k):
    world, poses, K, w, h = _render_job
    return world.render(poses[k], K, w, h)


def make_sequence(seq_id, n_frames, w=KITTI_W, h=KITTI_H, K=KITTI_K, workers=1):
    """Returns (frames uint8 (n,h,w), poses (n,3,4) camera-to-world, K).

    workers > 1 renders the frames in forked worker processes (same pixels; call it before the
    process touches the GPU).  With VO_SYNTH_CACHE=<dir> the rendered frames are kept there as .npy and
    loaded on the next call: profiling runs (rocprofv3 preloads a library that initialises the GPU
    before main, after which forking is not safe) render nothing and fork nothing."""
    global _render_job
    poses = trajectory(seq_id, n_frames)
    cache = os.environ.get("VO_SYNTH_CACHE")
    path = None
    if cache:
        path = os.path.join(cache, "seq%d_%d_%dx%d_%x.npy" % (seq_id, n_frames, w, h,
                                                                 abs(hash(tuple(np.round(K, 6).reshape(-1)))) & 0xffffff))
        if os.path.exists(path):
            return np.load(path), poses, K.copy()
    world = World(seq_id)
    if workers > 1 and n_frames > 8 and os.environ.get("VO_SYNTH_NOFORK", "0") != "1":
        import multiprocessing as mp

        _render_job = (world, poses, K, w, h)
        pool = mp.get_context("fork").Pool(workers)
        try:
            frames = np.stack(pool.map(_render_one, range(n_frames), chunksize=4))
        finally:
            pool.close()  # workers exit by themselves: no SIGTERM (terminate()) for a profiler to report
            pool.join()
        _render_job = None
    else:
        frames = np.stack([world.render(poses[k], K, w, h) for k in range(n_frames)])
    if path:
        os.makedirs(cache, exist_ok=True)
        tmp = path + ".%d.tmp.npy" % os.getpid()
        np.save(tmp, frames)
        os.replace(tmp, path)
    return frames, poses, K.copy()


def pingpong_indices(n_base, n_frames):
    """Frame k of a long sequence cut from n_base rendered frames: 0, 1, .., n_base-1, n_base-2, .., 1, 0,
    1, ..  Every consecutive pair is a pair of neighbouring base frames (forward or backward motion), so
    a sequence of any length has the statistics of the rendered one without rendering it whole."""
    period = 2 * (n_base - 1)
    k = np.arange(n_frames) % period
    return np.where(k < n_base, k, period - k).astype(np.int64)


def write_kitti_layout(root, seq_id, frames, poses, K=KITTI_K):
    """sequences/NN/image_0/%06d.png, calib.txt, times.txt, poses/NN.txt (reference README.md:54-94)."""
    from PIL import Image

    seq = "%02d" % seq_id
    img_dir = os.path.join(root, "sequences", seq, "image_0")
    os.makedirs(img_dir, exist_ok=True)
    os.makedirs(os.path.join(root, "poses"), exist_ok=True)
    for k, f in enumerate(frames):
        Image.fromarray(f, mode="L").save(os.path.join(img_dir, "%06d.png" % k))
    Pline = "%r %r %r 0 %r %r %r 0 0 0 1 0" % (float(K[0, 0]), 0.0, float(K[0, 2]), 0.0, float(K[1, 1]), float(K[1, 2]))
    with open(os.path.join(root, "sequences", seq, "calib.txt"), "w") as fh:
        for name in ("P0", "P1", "P2", "P3"):
            fh.write("%s: %s\n" % (name, Pline))
    with open(os.path.join(root, "sequences", seq, "times.txt"), "w") as fh:
        for k in range(len(frames)):
            fh.write("%e\n" % (0.1 * k))
    with open(os.path.join(root, "poses", seq + ".txt"), "w") as fh:
        for p in poses:
            fh.write(" ".join("%e" % x for x in p.reshape(-1)) + "\n")


def load_kitti_calibration(path, line_number=0):
    """loadKittiCalibration: line N, strip the 4-char 'P0: ' tag, 12 doubles -> 3x3 K
    (reference: src/kitti/kitti.cpp:10-38)."""
    with open(path) as fh:
        lines = fh.read().splitlines()
    vals = [float(x) for x in lines[line_number][4:].split(" ") if x]
    P = np.array(vals[:12]).reshape(3, 4)
    return P[:, :3].copy()


# ---- micro-benchmark inputs ------------------------------------------------

def matcher_descriptors(n, seed_q=1, seed_t=2, flip_p=0.08, planted=0.7):
    """Query bits iid uniform; train[j] = query[perm(j)] with Binomial(256, flip_p) bit flips for
    `planted` of the rows and fresh random rows for the rest."""
    rq = np.random.default_rng(seed_q)
    rt = np.random.default_rng(seed_t)
    q = rq.integers(0, 256, (n, 32), dtype=np.uint8)
    perm = rt.permutation(n)
    t = q[perm].copy()
    bits = np.unpackbits(t, axis=1)
    flips = rt.random(bits.shape) < flip_p
    bits ^= flips.astype(np.uint8)
    t = np.packbits(bits, axis=1)
    fresh = rt.random(n) >= planted
    t[fresh] = rt.integers(0, 256, (int(fresh.sum()), 32), dtype=np.uint8)
    return q, t


def rotation(axis_angle):
    a = np.asarray(axis_angle, float)
    th = np.linalg.norm(a)
    if th < 1e-15:
        return np.eye(3)
    k = a / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


def correspondences(n, seed=0, inlier_ratio=0.6, sigma=0.5, K=KITTI_K, w=KITTI_W, h=KITTI_H):
    """Two-view point matches with known (R, t): inliers are projections of random 3-D points with
    Gaussian pixel noise, outliers are uniform over the image.  x2 = R x1 + t (cv convention).
    Returns (p0 float32 (n,2), p1 float32 (n,2), R, t_unit, inlier_flags)."""
    rng = np.random.default_rng(seed)
    R = rotation(rng.normal(0, 0.02, 3))
    t = np.array([rng.normal(0, 0.05), rng.normal(0, 0.03), -1.0])
    t /= np.linalg.norm(t)
    p0 = np.zeros((n, 2))
    p1 = np.zeros((n, 2))
    got = 0
    while got < n:
        m = 2 * (n - got) + 16
        uv = np.stack([rng.uniform(0, w, m), rng.uniform(0, h, m)], 1)
        depth = rng.uniform(4.0, 60.0, m)
        X = np.stack([(uv[:, 0] - K[0, 2]) / K[0, 0] * depth, (uv[:, 1] - K[1, 2]) / K[1, 1] * depth, depth], 1)
        X2 = X @ R.T + t
        ok = X2[:, 2] > 1.0
        u2 = K[0, 0] * X2[:, 0] / X2[:, 2] + K[0, 2]
        v2 = K[1, 1] * X2[:, 1] / X2[:, 2] + K[1, 2]
        ok &= (u2 >= 0) & (u2 < w) & (v2 >= 0) & (v2 < h)
        k = min(int(ok.sum()), n - got)
        p0[got:got + k] = uv[ok][:k]
        p1[got:got + k] = np.stack([u2[ok][:k], v2[ok][:k]], 1)
        got += k
    p0 += rng.normal(0, sigma, p0.shape)
    p1 += rng.normal(0, sigma, p1.shape)
    inl = rng.random(n) < inlier_ratio
    n_out = int((~inl).sum())
    p1[~inl] = np.stack([rng.uniform(0, w, n_out), rng.uniform(0, h, n_out)], 1)
    return p0.astype(np.float32), p1.astype(np.float32), R, t, inl
