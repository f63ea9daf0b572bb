"""ctypes binding of oracle/libvo_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package (visual_odometry_amd/) never does.  See oracle/vo_oracle.h for what each call
restates and which reference line it follows.  Parity unpinned (no OpenCV in this image).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvo_oracle.so")

KP_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
     ("octave", "<i4"), ("class_id", "<i4")]
)
DM_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])
assert KP_DTYPE.itemsize == 28 and DM_DTYPE.itemsize == 16

LMEDS, RANSAC = 4, 8


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("edge_threshold", C.c_int32), ("first_level", C.c_int32), ("wta_k", C.c_int32),
                ("score_type", C.c_int32), ("patch_size", C.c_int32), ("fast_threshold", C.c_int32)]


class LkParams(C.Structure):
    _fields_ = [("win_w", C.c_int32), ("win_h", C.c_int32), ("max_level", C.c_int32), ("max_count", C.c_int32),
                ("epsilon", C.c_double), ("min_eig_threshold", C.c_double), ("channels", C.c_int32),
                ("reserved", C.c_int32)]


class AkazeParams(C.Structure):
    _fields_ = [("threshold", C.c_float), ("n_octaves", C.c_int32), ("n_sublevels", C.c_int32),
                ("channels", C.c_int32), ("pattern_size", C.c_int32), ("soffset", C.c_float),
                ("derivative_factor", C.c_float), ("kcontrast_percentile", C.c_float),
                ("kcontrast_nbins", C.c_int32), ("reserved", C.c_int32)]


def akaze_params(**kw):
    p = AkazeParams(0.001, 4, 4, 3, 10, 1.6, 1.5, 0.7, 300, 0)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def lk_params(**kw):
    p = LkParams(21, 21, 3, 30, 0.01, 1e-4, 3, 0)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def orb_params(nfeatures=2000, **kw):
    p = OrbParams(nfeatures, 1.2, 8, 31, 0, 2, 0, 31, 20)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def build():
    """Compile the oracle with the recipe in oracle/Makefile (gcc only)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
    return _lib


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t) if a is not None else None


def _img(gray):
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    assert gray.ndim == 2
    return gray


def fast_detect(gray, threshold=20, nms=True):
    gray = _img(gray)
    h, w = gray.shape
    cap = w * h
    out = np.zeros(cap, KP_DTYPE)
    n = C.c_int(0)
    rc = lib().vo_oracle_fast_detect(_p(gray), w, h, w, int(threshold), int(bool(nms)), _p(out), cap,
                                     C.byref(n))
    assert rc == 0, rc
    return out[: n.value].copy()


def resize_linear_exact(src, dw, dh):
    src = _img(src)
    sh, sw = src.shape
    dst = np.zeros((dh, dw), np.uint8)
    rc = lib().vo_oracle_resize_linear_exact(_p(src), sw, sh, sw, _p(dst), dw, dh, dw)
    assert rc == 0
    return dst


def gauss7(src):
    src = _img(src)
    h, w = src.shape
    dst = np.zeros_like(src)
    assert lib().vo_oracle_gauss7(_p(src), w, h, w, _p(dst), w) == 0
    return dst


def orb_layout(w, h, params):
    lw = (C.c_int * 16)()
    lh = (C.c_int * 16)()
    ls = (C.c_float * 16)()
    q = (C.c_int * 16)()
    n = lib().vo_oracle_orb_layout(w, h, C.byref(params), lw, lh, ls, q)
    assert n > 0
    return [(lw[i], lh[i], ls[i], q[i]) for i in range(n)]


def orb_level(gray, params, level, blurred=False):
    gray = _img(gray)
    h, w = gray.shape
    lw, lh, _, _ = orb_layout(w, h, params)[level]
    out = np.zeros((lh, lw), np.uint8)
    rc = lib().vo_oracle_orb_level(_p(gray), w, h, w, C.byref(params), level, int(blurred), _p(out))
    assert rc == 0
    return out


def orb_candidates(gray, params, level):
    gray = _img(gray)
    h, w = gray.shape
    cap = w * h // 4
    xs = np.zeros(cap, np.int32)
    ys = np.zeros(cap, np.int32)
    fs = np.zeros(cap, np.float32)
    hs = np.zeros(cap, np.float32)
    n = C.c_int(0)
    rc = lib().vo_oracle_orb_candidates(_p(gray), w, h, w, C.byref(params), level, _p(xs), _p(ys),
                                        _p(fs), _p(hs), cap, C.byref(n))
    assert rc == 0
    k = n.value
    return xs[:k].copy(), ys[:k].copy(), fs[:k].copy(), hs[:k].copy()


def orb_detect_compute(gray, params):
    gray = _img(gray)
    h, w = gray.shape
    cap = max(4 * params.nfeatures, 64)
    kp = np.zeros(cap, KP_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    n = C.c_int(0)
    rc = lib().vo_oracle_orb_detect_compute(_p(gray), w, h, w, C.byref(params), _p(kp), _p(desc), cap,
                                            C.byref(n))
    assert rc == 0, rc
    return kp[: n.value].copy(), desc[: n.value].copy()


def orb_compute(gray, params, kps):
    gray = _img(gray)
    h, w = gray.shape
    kp = np.array(kps, dtype=KP_DTYPE, copy=True)
    n = C.c_int(len(kp))
    desc = np.zeros((max(len(kp), 1), 32), np.uint8)
    rc = lib().vo_oracle_orb_compute(_p(gray), w, h, w, C.byref(params), _p(kp), C.byref(n), _p(desc))
    assert rc == 0, rc
    return kp[: n.value].copy(), desc[: n.value].copy()


def knn2_hamming(q, t):
    q = np.ascontiguousarray(q, np.uint8)
    t = np.ascontiguousarray(t, np.uint8)
    nq, nt = q.shape[0], t.shape[0]
    d = q.shape[1] if q.ndim == 2 and nq else (t.shape[1] if t.ndim == 2 and nt else 32)
    out = np.zeros((nq, 2), DM_DTYPE)
    nv = np.zeros(nq, np.int32)
    rc = lib().vo_oracle_knn2_hamming(_p(q), nq, _p(t), nt, d, _p(out), _p(nv))
    assert rc == 0
    return out, nv


def match_filter(knn, n_valid, kq, kt, ratio=0.7, max_disp=200.0):
    knn = np.ascontiguousarray(knn, DM_DTYPE)
    nq = knn.shape[0]
    good = np.zeros(max(nq, 1), DM_DTYPE)
    n = C.c_int(0)
    kq = np.ascontiguousarray(kq, KP_DTYPE)
    kt = np.ascontiguousarray(kt, KP_DTYPE)
    nv = np.ascontiguousarray(n_valid, np.int32) if n_valid is not None else None
    rc = lib().vo_oracle_match_filter(_p(knn), _p(nv), nq, _p(kq), _p(kt), C.c_float(ratio),
                                      C.c_float(max_disp), _p(good), C.byref(n))
    assert rc == 0
    return good[: n.value].copy()


def rng_next(seed, n):
    out = np.zeros(n, np.uint32)
    lib().vo_oracle_rng_next(C.c_uint64(seed & (2**64 - 1)), n, _p(out))
    return out


def sample_subsets(seed, npoints, niters):
    idx = np.zeros((niters, 5), np.int32)
    rc = lib().vo_oracle_sample_subsets(C.c_uint64(seed & (2**64 - 1)), npoints, niters, _p(idx))
    assert rc == 0
    return idx


RETAIN_CV342, RETAIN_LATER = 0, 1


def set_retain_best_split(mode):
    """KeyPointsFilter::retainBest's nth_element position: RETAIN_CV342 (begin + n, the reference's OpenCV
    3.4.2, the default) or RETAIN_LATER (begin + n - 1, later OpenCV releases).  Process-wide."""
    lib().vo_oracle_set_retain_best_split(int(mode))


def get_retain_best_split():
    return lib().vo_oracle_get_retain_best_split()


def five_point(x1, x2):
    x1 = np.ascontiguousarray(x1, np.float64)
    x2 = np.ascontiguousarray(x2, np.float64)
    E = np.zeros((10, 3, 3))
    n = lib().vo_oracle_five_point(_p(x1), _p(x2), _p(E))
    return E[:n].copy()


def sampson_err(E, x1, x2):
    E = np.ascontiguousarray(E, np.float64)
    x1 = np.ascontiguousarray(x1, np.float64)
    x2 = np.ascontiguousarray(x2, np.float64)
    n = x1.shape[0]
    err = np.zeros(n, np.float32)
    lib().vo_oracle_sampson_err(_p(E), _p(x1), _p(x2), n, _p(err))
    return err


def find_essential(p0, p1, K, method=RANSAC, prob=0.999, threshold=1.0, seed=2**64 - 1, max_iters=1000):
    """Returns (rc, E, mask, stats)."""
    p0 = np.ascontiguousarray(p0, np.float32)
    p1 = np.ascontiguousarray(p1, np.float32)
    K = np.ascontiguousarray(K, np.float64)
    n = p0.shape[0]
    E = np.zeros((3, 3))
    mask = np.zeros(max(n, 1), np.uint8)
    stats = np.zeros(4, np.int32)
    rc = lib().vo_oracle_find_essential(_p(p0), _p(p1), n, _p(K), method, C.c_double(prob),
                                        C.c_double(threshold), C.c_uint64(seed & (2**64 - 1)),
                                        max_iters, _p(E), _p(mask), _p(stats))
    return rc, E, mask[:n].copy(), stats


def decompose_essential(E):
    E = np.ascontiguousarray(E, np.float64)
    R1 = np.zeros((3, 3))
    R2 = np.zeros((3, 3))
    t = np.zeros(3)
    lib().vo_oracle_decompose_essential(_p(E), _p(R1), _p(R2), _p(t))
    return R1, R2, t


def triangulate_points(P0, P1, x0, x1):
    P0 = np.ascontiguousarray(P0, np.float64)
    P1 = np.ascontiguousarray(P1, np.float64)
    x0 = np.ascontiguousarray(x0, np.float64)
    x1 = np.ascontiguousarray(x1, np.float64)
    n = x0.shape[0]
    out = np.zeros((4, n))
    assert lib().vo_oracle_triangulate_points(_p(P0), _p(P1), _p(x0), _p(x1), n, _p(out)) == 0
    return out


def recover_pose(E, p0, p1, K, dist_thresh=200.0, mask=None):
    """Returns (n_good, R, t, mask, pts4d)."""
    E = np.ascontiguousarray(E, np.float64)
    p0 = np.ascontiguousarray(p0, np.float32)
    p1 = np.ascontiguousarray(p1, np.float32)
    K = np.ascontiguousarray(K, np.float64)
    n = p0.shape[0]
    m = np.array(mask, np.uint8, copy=True) if mask is not None else None
    R = np.zeros((3, 3))
    t = np.zeros(3)
    pts = np.zeros((4, max(n, 1)))
    ng = C.c_int(0)
    rc = lib().vo_oracle_recover_pose(_p(E), _p(p0), _p(p1), n, _p(K), C.c_double(dist_thresh), _p(m),
                                      _p(R), _p(t), _p(pts), C.byref(ng))
    assert rc == 0
    return ng.value, R, t, m, pts[:, :n].copy()


def triangulate(p0, p1, P0, P1):
    p0 = np.ascontiguousarray(p0, np.float32)
    p1 = np.ascontiguousarray(p1, np.float32)
    P0 = np.ascontiguousarray(P0, np.float64)
    P1 = np.ascontiguousarray(P1, np.float64)
    n = p0.shape[0]
    out = np.zeros((n, 3))
    assert lib().vo_oracle_triangulate(_p(p0), _p(p1), n, _p(P0), _p(P1), _p(out)) == 0
    return out


def projection_matrix(K, pose):
    K = np.ascontiguousarray(K, np.float64)
    pose = np.ascontiguousarray(pose, np.float64)
    P = np.zeros((3, 4))
    lib().vo_oracle_projection_matrix(_p(K), _p(pose), _p(P))
    return P


# ---- tracker -------------------------------------------------------------------------------
def lk_levels(w, h, params):
    lw = (C.c_int * 16)()
    lh = (C.c_int * 16)()
    n = lib().vo_oracle_lk_levels(int(w), int(h), C.byref(params), lw, lh)
    return [(lw[i], lh[i]) for i in range(n)]


def lk_level(gray, params, level):
    """(image, deriv) of one tracker pyramid level; deriv is h x w x 2 int16 (dx, dy)."""
    gray = _img(gray)
    h, w = gray.shape
    lw, lh = lk_levels(w, h, params)[level]
    img = np.zeros((lh, lw), np.uint8)
    der = np.zeros((lh, lw, 2), np.int16)
    rc = lib().vo_oracle_lk_level(_p(gray), w, h, w, C.byref(params), int(level), _p(img), _p(der))
    assert rc == 0, rc
    return img, der


def lk_track(prev, nxt, prev_pts, params):
    prev, nxt = _img(prev), _img(nxt)
    h, w = prev.shape
    assert nxt.shape == prev.shape
    p0 = np.ascontiguousarray(prev_pts, dtype=np.float32).reshape(-1, 2)
    n = len(p0)
    p1 = np.zeros((max(n, 1), 2), np.float32)
    st = np.zeros(max(n, 1), np.uint8)
    rc = lib().vo_oracle_lk_track(_p(prev), _p(nxt), w, h, w, _p(p0), n, C.byref(params), _p(p1), _p(st))
    assert rc == 0, rc
    return p1[:n].copy(), st[:n].copy()


def remove_points(p0, p1, status, pts3d=None, mask=None):
    """removePoints(): returns (p0, p1, status, pts3d, mask, ret)."""
    p0 = np.array(p0, dtype=np.float32, copy=True).reshape(-1, 2)
    p1 = np.array(p1, dtype=np.float32, copy=True).reshape(-1, 2)
    st = np.array(status, dtype=np.uint8, copy=True)
    n = len(p0)
    p3 = np.array(pts3d if pts3d is not None else np.zeros((0, 3)), dtype=np.float64, copy=True).reshape(-1, 3)
    mk = np.array(mask if mask is not None else np.zeros(0), dtype=np.uint8, copy=True).reshape(-1)
    n_out, n3, nm, ret = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    rc = lib().vo_oracle_remove_points(_p(p0), _p(p1), n, _p(st), _p(p3), len(p3), _p(mk), len(mk), C.byref(n_out),
                                       C.byref(n3), C.byref(nm), C.byref(ret))
    assert rc == 0, rc
    return p0[: n_out.value].copy(), p1[: n_out.value].copy(), st, p3[: n3.value].copy(), mk[: nm.value].copy(), ret.value


# ---- AKAZE ---------------------------------------------------------------------------------------
def akaze_layout(w, h, params):
    cap = 64
    lw, lh, ss, bd, ns = (np.zeros(cap, np.int32) for _ in range(5))
    es = np.zeros(cap, np.float32)
    n = lib().vo_oracle_akaze_layout(int(w), int(h), C.byref(params), _p(lw), _p(lh), _p(ss), _p(bd), _p(es), _p(ns))
    assert n > 0, n
    return dict(w=lw[:n].copy(), h=lh[:n].copy(), sigma_size=ss[:n].copy(), border=bd[:n].copy(),
                esigma=es[:n].copy(), nsteps=ns[:n].copy())


def akaze_tau(T):
    tau = np.zeros(256, np.float32)
    lib().vo_oracle_akaze_tau.argtypes = [C.c_float, C.c_void_p, C.c_int]
    n = lib().vo_oracle_akaze_tau(float(T), _p(tau), 256)
    return tau[:n].copy()


def akaze_level(gray, params, level, which):
    gray = _img(gray)
    h, w = gray.shape
    lay = akaze_layout(w, h, params)
    out = np.zeros((int(lay["h"][level]), int(lay["w"][level])), np.float32)
    rc = lib().vo_oracle_akaze_level(_p(gray), w, h, w, C.byref(params), int(level), int(which), _p(out))
    assert rc == 0, rc
    return out


def akaze_detect_compute(gray, params, cap=65536):
    gray = _img(gray)
    h, w = gray.shape
    kp = np.zeros(cap, KP_DTYPE)
    desc = np.zeros((cap, 61), np.uint8)
    n = C.c_int(0)
    rc = lib().vo_oracle_akaze_detect_compute(_p(gray), w, h, w, C.byref(params), _p(kp), _p(desc), cap, C.byref(n))
    assert rc == 0, rc
    return kp[: n.value].copy(), desc[: n.value].copy()
