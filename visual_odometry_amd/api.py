"""numpy-level binding of the C ABI (include/vo_hip.h): one method per entry point.

This is plumbing for tests and bench.py; the host-side mirror of the reference's C++ interface
lives in visual_odometry_amd/host/ (C++) and visual_odometry_amd/reference_api.py.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import OrbParams, Limits, PairParams, PairResult, LkParams, SeqOutputs, AkazeParams

KP_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
     ("octave", "<i4"), ("class_id", "<i4")]
)
DM_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])

LMEDS, RANSAC = 4, 8
OPT_SEL_CAP, OPT_LAST_FLAGS, OPT_CAND_CAP = 0, 1, 2
OPT_KNN_MFMA, OPT_ORB_SPLIT, OPT_SEQ_PREFETCH, OPT_AKAZE_GRAPH, OPT_SOLVER, OPT_KNN_SPLITS, OPT_LK_GENERIC = 3, 4, 5, 6, 7, 8, 9
VO_OK, VO_ERR_INVALID, VO_ERR_CAPACITY, VO_ERR_DEVICE, VO_ERR_FAILED = 0, 1, 2, 3, 4


class VoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("vo_hip error %d: %s" % (code, msg))
        self.code = code


def orb_params(nfeatures=2000, **kw):
    p = OrbParams()
    _lib.lib().vo_orb_params_default(C.byref(p), int(nfeatures))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def akaze_params(**kw):
    p = AkazeParams()
    _lib.lib().vo_akaze_params_default(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def lk_params(**kw):
    p = LkParams()
    _lib.lib().vo_lk_params_default(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def pair_params(nfeatures=2000, **kw):
    p = PairParams()
    _lib.lib().vo_pair_params_default(C.byref(p), int(nfeatures))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _img(gray):
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    assert gray.ndim == 2
    return gray


class VoContext:
    """One vo_ctx (one HIP stream + workspace) on `device`."""

    def __init__(self, device=0, **limits):
        self.L = _lib.lib()
        lim = Limits()
        self.L.vo_limits_default(C.byref(lim))
        for k, v in limits.items():
            setattr(lim, k, v)
        self.limits = lim
        h = C.c_void_p()
        rc = self.L.vo_ctx_create(int(device), C.byref(lim), C.byref(h))
        if rc != 0:
            raise VoError(rc, "vo_ctx_create failed (no usable HIP device? there is no CPU fallback)")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.vo_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, allow=()):
        if rc != 0 and rc not in allow:
            raise VoError(rc, self.L.vo_last_error(self.h).decode())
        return rc

    def synchronize(self):
        self._check(self.L.vo_ctx_synchronize(self.h))

    # ---- tracker -----------------------------------------------------------------------
    def lk_track(self, prev, nxt, prev_pts, params):
        """prev may be None: track from the previous call's `nxt` image (its pyramid is cached)."""
        nxt = _img(nxt)
        h, w = nxt.shape
        stride = nxt.strides[0]
        if prev is not None:
            prev = _img(prev)
            assert prev.shape == nxt.shape and prev.strides[0] == stride
        p0 = np.ascontiguousarray(prev_pts, dtype=np.float32).reshape(-1, 2)
        n = len(p0)
        p1 = np.zeros((max(n, 1), 2), np.float32)
        st = np.zeros(max(n, 1), np.uint8)
        self._check(self.L.vo_lk_track(self.h, _p(prev), _p(nxt), w, h, stride, _p(p0), n, C.byref(params), _p(p1),
                                       _p(st)))
        return p1[:n].copy(), st[:n].copy()

    def lk_track_batch(self, prevs, nexts, pts, params):
        """S image pairs in one submission (vo_lk_track_batch).  prevs: list of images or None (track from the
        previous call's `nexts`); pts: list of (n_s, 2) arrays.  Returns lists (next points, status)."""
        S = len(nexts)
        nexts = [_img(a) for a in nexts]
        h, w = nexts[0].shape
        stride = nexts[0].strides[0]
        pv = None
        if prevs is not None:
            prevs = [_img(a) for a in prevs]
            pv = (C.c_void_p * S)(*[a.ctypes.data for a in prevs])
        nx = (C.c_void_p * S)(*[a.ctypes.data for a in nexts])
        p0 = [np.ascontiguousarray(p, dtype=np.float32).reshape(-1, 2) for p in pts]
        n = (C.c_int * S)(*[len(p) for p in p0])
        p1 = [np.zeros((max(len(p), 1), 2), np.float32) for p in p0]
        st = [np.zeros(max(len(p), 1), np.uint8) for p in p0]
        a0 = (C.c_void_p * S)(*[p.ctypes.data for p in p0])
        a1 = (C.c_void_p * S)(*[p.ctypes.data for p in p1])
        a2 = (C.c_void_p * S)(*[p.ctypes.data for p in st])
        self._check(self.L.vo_lk_track_batch(self.h, S, pv, nx, w, h, stride, a0, n, C.byref(params), a1, a2))
        return [p1[s][: len(p0[s])].copy() for s in range(S)], [st[s][: len(p0[s])].copy() for s in range(S)]

    def essential_pose_batch(self, p0s, p1s, K, method=RANSAC, prob=0.999, threshold=1.0, seed=2**64 - 1,
                             max_iters=1000, dist_thresh=200.0):
        """vo_essential_pose_batch: list of per-set tuples (rc, E, n_good, R, t, mask, pts4d)."""
        S = len(p0s)
        p0 = [np.ascontiguousarray(p, np.float32).reshape(-1, 2) for p in p0s]
        p1 = [np.ascontiguousarray(p, np.float32).reshape(-1, 2) for p in p1s]
        K = np.ascontiguousarray(K, np.float64)
        n = (C.c_int * S)(*[len(p) for p in p0])
        E = np.zeros((S, 3, 3))
        R = np.zeros((S, 3, 3))
        t = np.zeros((S, 3))
        mask = [np.zeros(max(len(p), 1), np.uint8) for p in p0]
        pts = [np.zeros((4, max(len(p), 1))) if len(p) == 0 else np.zeros((4, len(p))) for p in p0]
        ng = (C.c_int * S)()
        stt = (C.c_int * S)()
        a0 = (C.c_void_p * S)(*[p.ctypes.data for p in p0])
        a1 = (C.c_void_p * S)(*[p.ctypes.data for p in p1])
        am = (C.c_void_p * S)(*[p.ctypes.data for p in mask])
        ap = (C.c_void_p * S)(*[p.ctypes.data for p in pts])
        self._check(self.L.vo_essential_pose_batch(self.h, S, a0, a1, n, _p(K), int(method), C.c_double(prob),
                                                   C.c_double(threshold), C.c_uint64(seed & (2**64 - 1)), int(max_iters),
                                                   C.c_double(dist_thresh), _p(E), am, _p(R), _p(t), ap, ng, stt))
        return [(stt[s], E[s].copy(), ng[s], R[s].copy(), t[s].copy(), mask[s][: len(p0[s])].copy(),
                 pts[s][:, : len(p0[s])].copy()) for s in range(S)]

    def debug_read_lk_level(self, which, level):
        """which: 0 prev pyramid, 1 next pyramid, 2 derivative image (h x w x 2 int16)."""
        w, h = C.c_int(0), C.c_int(0)
        cap = self.limits.max_width * self.limits.max_height * 4
        buf = np.zeros(cap, np.uint8)
        self._check(self.L.vo_debug_read_lk_level(self.h, int(which), int(level), _p(buf), cap, C.byref(w), C.byref(h)))
        if which == 2:
            return buf[: w.value * h.value * 4].view(np.int16).reshape(h.value, w.value, 2).copy()
        return buf[: w.value * h.value].reshape(h.value, w.value).copy()

    # ---- features ----------------------------------------------------------------------
    def fast_detect(self, gray, threshold=20, nms=True, cap=None):
        gray = _img(gray)
        h, w = gray.shape
        cap = w * h // 4 if cap is None else cap
        out = np.zeros(max(cap, 1), KP_DTYPE)
        n = C.c_int(0)
        self._check(self.L.vo_fast_detect(self.h, _p(gray), w, h, gray.strides[0], int(threshold),
                                          int(bool(nms)), _p(out), cap, C.byref(n)))
        return out[: n.value].copy()

    def orb_detect_compute(self, gray, params, cap=None):
        gray = _img(gray)
        h, w = gray.shape
        cap = max(2 * params.nfeatures, 64) if cap is None else cap
        kp = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        self._check(self.L.vo_orb_detect_compute(self.h, _p(gray), w, h, gray.strides[0], C.byref(params),
                                                 _p(kp), _p(desc), cap, C.byref(n)))
        return kp[: n.value].copy(), desc[: n.value].copy()

    def orb_compute(self, gray, params, kps):
        gray = _img(gray)
        h, w = gray.shape
        kp = np.array(kps, dtype=KP_DTYPE, copy=True)
        n = C.c_int(len(kp))
        desc = np.zeros((max(len(kp), 1), 32), np.uint8)
        self._check(self.L.vo_orb_compute(self.h, _p(gray), w, h, gray.strides[0], C.byref(params), _p(kp),
                                          C.byref(n), _p(desc)))
        return kp[: n.value].copy(), desc[: n.value].copy()

    # ---- matcher -------------------------------------------------------------------------
    def knn2_hamming(self, q, t):
        q = np.ascontiguousarray(q, np.uint8)
        t = np.ascontiguousarray(t, np.uint8)
        nq, nt = q.shape[0], t.shape[0]
        d = q.shape[1] if q.ndim == 2 and nq else (t.shape[1] if t.ndim == 2 and nt else 32)
        out = np.zeros((nq, 2), DM_DTYPE)
        nv = np.zeros(max(nq, 1), np.int32)
        self._check(self.L.vo_knn2_hamming(self.h, _p(q), nq, _p(t), nt, d, _p(out), _p(nv)))
        return out, nv[:nq]

    def knn2_hamming_cross(self, q, t):
        """knn2_hamming plus rev_nn[j] = nearest query of train row j (one launch, both directions)."""
        q = np.ascontiguousarray(q, np.uint8)
        t = np.ascontiguousarray(t, np.uint8)
        nq, nt = q.shape[0], t.shape[0]
        d = q.shape[1] if q.ndim == 2 and nq else (t.shape[1] if t.ndim == 2 and nt else 32)
        out = np.zeros((nq, 2), DM_DTYPE)
        nv = np.zeros(max(nq, 1), np.int32)
        rev = np.zeros(max(nt, 1), np.int32)
        self._check(self.L.vo_knn2_hamming_cross(self.h, _p(q), nq, _p(t), nt, d, _p(out), _p(nv), _p(rev)))
        return out, nv[:nq], rev[:nt]

    def knn2_hamming_static(self, d_q, nq, d_t, nt, desc_bytes, d_out):
        """Device addresses in and out, no allocation, asynchronous on the context's stream."""
        self._check(self.L.vo_knn2_hamming_static(self.h, C.c_void_p(int(d_q)), int(nq), C.c_void_p(int(d_t)), int(nt),
                                                  int(desc_bytes), C.c_void_p(int(d_out))))

    def match_filter(self, knn, n_valid, kq, kt, ratio=0.7, max_disp=200.0, rev_nn=None):
        knn = np.ascontiguousarray(knn, DM_DTYPE)
        nq = knn.shape[0]
        good = np.zeros(max(nq, 1), DM_DTYPE)
        n = C.c_int(0)
        kq = np.ascontiguousarray(kq, KP_DTYPE)
        kt = np.ascontiguousarray(kt, KP_DTYPE)
        nv = np.ascontiguousarray(n_valid, np.int32) if n_valid is not None else None
        if rev_nn is not None:
            rev = np.ascontiguousarray(rev_nn, np.int32)
            self._check(self.L.vo_match_filter_cross(self.h, _p(knn), _p(nv), nq, _p(rev), len(rev), _p(kq), _p(kt),
                                                     C.c_float(ratio), C.c_float(max_disp), _p(good), C.byref(n)))
        else:
            self._check(self.L.vo_match_filter(self.h, _p(knn), _p(nv), nq, _p(kq), _p(kt), C.c_float(ratio),
                                               C.c_float(max_disp), _p(good), C.byref(n)))
        return good[: n.value].copy()

    # ---- sfm -----------------------------------------------------------------------------
    def find_essential(self, p0, p1, K, method=RANSAC, prob=0.999, threshold=1.0, seed=2**64 - 1,
                       max_iters=1000):
        """Returns (rc, E, mask); rc is VO_OK or VO_ERR_FAILED (cv's empty-matrix case)."""
        p0 = np.ascontiguousarray(p0, np.float32).reshape(-1, 2)
        p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
        K = np.ascontiguousarray(K, np.float64)
        n = p0.shape[0]
        E = np.zeros((3, 3))
        mask = np.zeros(max(n, 1), np.uint8)
        rc = self.L.vo_find_essential(self.h, _p(p0), _p(p1), n, _p(K), int(method), C.c_double(prob),
                                      C.c_double(threshold), C.c_uint64(seed & (2**64 - 1)), int(max_iters),
                                      _p(E), _p(mask))
        self._check(rc, allow=(VO_ERR_FAILED,))
        return rc, E, mask[:n].copy()

    def recover_pose(self, E, p0, p1, K, dist_thresh=200.0, mask=None):
        """Returns (n_good, R, t, mask, pts4d)."""
        E = np.ascontiguousarray(E, np.float64)
        p0 = np.ascontiguousarray(p0, np.float32).reshape(-1, 2)
        p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
        K = np.ascontiguousarray(K, np.float64)
        n = p0.shape[0]
        m = np.array(mask, np.uint8, copy=True) if mask is not None else None
        R = np.zeros((3, 3))
        t = np.zeros(3)
        pts = np.zeros((4, max(n, 1)))
        if n > 0:
            pts = np.zeros((4, n))
        ng = C.c_int(0)
        self._check(self.L.vo_recover_pose(self.h, _p(E), _p(p0), _p(p1), n, _p(K), C.c_double(dist_thresh),
                                           _p(m), _p(R), _p(t), _p(pts), C.byref(ng)))
        return ng.value, R, t, m, pts[:, :n].copy()

    def essential_pose(self, p0, p1, K, method=RANSAC, prob=0.999, threshold=1.0, seed=2**64 - 1, max_iters=1000,
                       dist_thresh=200.0):
        """find_essential + recover_pose in one submission. Returns (rc, E, n_good, R, t, mask, pts4d)."""
        p0 = np.ascontiguousarray(p0, np.float32).reshape(-1, 2)
        p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
        K = np.ascontiguousarray(K, np.float64)
        n = p0.shape[0]
        E = np.zeros((3, 3))
        R = np.zeros((3, 3))
        t = np.zeros(3)
        mask = np.zeros(max(n, 1), np.uint8)
        pts = np.zeros((4, max(n, 1)))
        if n > 0:
            pts = np.zeros((4, n))
        ng = C.c_int(0)
        rc = self.L.vo_essential_pose(self.h, _p(p0), _p(p1), n, _p(K), int(method), C.c_double(prob),
                                      C.c_double(threshold), C.c_uint64(seed & (2**64 - 1)), int(max_iters),
                                      C.c_double(dist_thresh), _p(E), _p(mask), _p(R), _p(t), _p(pts), C.byref(ng))
        self._check(rc, allow=(VO_ERR_FAILED,))
        return rc, E, ng.value, R, t, mask[:n].copy(), pts[:, :n].copy()

    def triangulate(self, p0, p1, P0, P1):
        p0 = np.ascontiguousarray(p0, np.float32).reshape(-1, 2)
        p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
        P0 = np.ascontiguousarray(P0, np.float64)
        P1 = np.ascontiguousarray(P1, np.float64)
        n = p0.shape[0]
        out = np.zeros((n, 3))
        self._check(self.L.vo_triangulate(self.h, _p(p0), _p(p1), n, _p(P0), _p(P1), _p(out)))
        return out

    def akaze_detect_compute(self, gray, params=None, cap=65536):
        """cv::AKAZE::create()->detectAndCompute: (keypoints, descriptors n x 61)."""
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        if params is None:
            params = akaze_params()
        kp = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 61), np.uint8)
        n = C.c_int(0)
        self._check(self.L.vo_akaze_detect_compute(self.h, _p(gray), w, h, w, C.byref(params), _p(kp), _p(desc), cap,
                                                   C.byref(n)))
        return kp[: n.value].copy(), desc[: n.value].copy()

    def debug_akaze_level(self, level, which):
        lw, lh = C.c_int(0), C.c_int(0)
        self._check(self.L.vo_debug_akaze_level(self.h, int(level), int(which), None, C.byref(lw), C.byref(lh)))
        out = np.zeros((lh.value, lw.value), np.float32)
        self._check(self.L.vo_debug_akaze_level(self.h, int(level), int(which), _p(out), None, None))
        return out

    def five_point(self, x1, x2, variant=1):
        """count x 5 x 2 normalised correspondences -> (models count x 10 x 3 x 3, nmodels count)."""
        x1 = np.ascontiguousarray(x1, np.float64).reshape(-1, 10)
        x2 = np.ascontiguousarray(x2, np.float64).reshape(-1, 10)
        n = x1.shape[0]
        models = np.zeros((n, 10, 3, 3))
        nm = np.zeros(n, np.int32)
        self._check(self.L.vo_five_point(self.h, _p(x1), _p(x2), n, int(variant), _p(models), _p(nm)))
        return models, nm

    def projection_matrix(self, K, pose):
        K = np.ascontiguousarray(K, np.float64)
        pose = np.ascontiguousarray(pose, np.float64)
        P = np.zeros((3, 4))
        self.L.vo_projection_matrix(_p(K), _p(pose), _p(P))
        return P

    # ---- device-resident batch -----------------------------------------------------------
    def pairs_run_dev(self, d_ptr, n_frames, w, h, stride, frame_stride, K, params):
        """d_ptr: integer device address of the frames. Returns a list of PairResult."""
        K = np.ascontiguousarray(K, np.float64)
        res = (PairResult * max(n_frames - 1, 1))()
        self._check(self.L.vo_pairs_run_dev(self.h, C.c_void_p(d_ptr), int(n_frames), int(w), int(h), int(stride),
                                            C.c_size_t(frame_stride), _p(K), C.byref(params), res))
        return res

    def pairs_get_frame(self, frame, cap=32768):
        kp = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        self._check(self.L.vo_pairs_get_frame(self.h, int(frame), _p(kp), _p(desc), cap, C.byref(n)))
        return kp[: n.value].copy(), desc[: n.value].copy()

    def pairs_get_matches(self, pair, cap=32768):
        good = np.zeros(cap, DM_DTYPE)
        m1 = np.zeros(cap, np.uint8)
        m2 = np.zeros(cap, np.uint8)
        n = C.c_int(0)
        self._check(self.L.vo_pairs_get_matches(self.h, int(pair), _p(good), _p(m1), _p(m2), cap, C.byref(n)))
        return good[: n.value].copy(), m1[: n.value].copy(), m2[: n.value].copy()

    # ---- knobs ---------------------------------------------------------------------------
    def set_option(self, option, value):
        self._check(self.L.vo_ctx_set_option(self.h, int(option), int(value)))

    def get_option(self, option):
        v = C.c_int(0)
        self._check(self.L.vo_ctx_get_option(self.h, int(option), C.byref(v)))
        return v.value

    def debug_read_level(self, which, frame, level):
        """Level `level` of the last ORB call's pyramid (which=0) or blurred pyramid (which=1)."""
        cap = self.limits.max_width * self.limits.max_height
        out = np.zeros(cap, np.uint8)
        w, h = C.c_int(0), C.c_int(0)
        self._check(self.L.vo_debug_read_level(self.h, int(which), int(frame), int(level), _p(out), cap,
                                               C.byref(w), C.byref(h)))
        return out[: w.value * h.value].reshape(h.value, w.value).copy()

    # ---- measurement hooks ---------------------------------------------------------------
    def prof_enable(self, on=True):
        self._check(self.L.vo_prof_enable(self.h, int(bool(on))))

    def prof_get(self, which, reset=True):
        n = C.c_int(0)
        ms = C.c_double(0)
        self._check(self.L.vo_prof_get(self.h, int(which), int(bool(reset)), C.byref(n), C.byref(ms)))
        return n.value, ms.value


MEM_HOST, MEM_DEVICE, MEM_HOST_PINNED = 0, 1, 2


class _BorrowedContext(VoContext):
    """A vo_ctx owned by a sequence runner (measurement hooks / options only)."""

    def __init__(self, lib, handle, limits):
        self.L = lib
        self.h = C.c_void_p(handle)
        self.limits = limits

    def close(self):
        self.h = None


class VoSequence:
    """vo_seq: a sequence pipelined over `n_streams` contexts in chunks of `pairs_per_chunk` pairs."""

    def __init__(self, device=0, n_streams=4, pairs_per_chunk=32, **limits):
        self.L = _lib.lib()
        lim = Limits()
        self.L.vo_limits_default(C.byref(lim))
        for k, v in limits.items():
            setattr(lim, k, v)
        self.limits = lim
        h = C.c_void_p()
        rc = self.L.vo_seq_create(int(device), C.byref(lim), int(n_streams), int(pairs_per_chunk), C.byref(h))
        if rc != 0:
            raise VoError(rc, "vo_seq_create failed (no usable HIP device? there is no CPU fallback)")
        self.h = h
        self.n_streams = n_streams
        self.pairs_per_chunk = pairs_per_chunk

    def close(self):
        if getattr(self, "h", None):
            self.L.vo_seq_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def ctx(self, i=0):
        return _BorrowedContext(self.L, self.L.vo_seq_ctx(self.h, int(i)), self.limits)

    def stats(self):
        """(chunks, host ms queueing, host ms waiting) of the last run."""
        n, a, b = C.c_int(0), C.c_double(0), C.c_double(0)
        self.L.vo_seq_get_stats(self.h, C.byref(n), C.byref(a), C.byref(b))
        return n.value, a.value, b.value

    def run_multi(self, seqs, w, h, stride, frame_stride, K, params, mem=MEM_DEVICE, outputs=None):
        """seqs: list of (address, n_frames); outputs: None or a list of (matches, inlier_mask, pose_mask) per
        sequence. Returns the list of result arrays. One pipeline over all chunks of all sequences."""
        n = len(seqs)
        K = np.ascontiguousarray(K, np.float64)
        ptrs = (C.c_void_p * n)(*[C.c_void_p(int(p)) for p, _ in seqs])
        counts = (C.c_int * n)(*[int(k) for _, k in seqs])
        res = [(PairResult * max(k - 1, 1))() for _, k in seqs]
        resp = (C.c_void_p * n)(*[C.cast(r, C.c_void_p) for r in res])
        outs = None
        if outputs is not None:
            outs = (SeqOutputs * n)()
            for i, (m, im, pm) in enumerate(outputs):
                cap = next(a.shape[1] for a in (m, im, pm) if a is not None)
                for a in (m, im, pm):
                    assert a is None or (a.flags.c_contiguous and a.shape == (seqs[i][1] - 1, cap))
                outs[i] = SeqOutputs(_p(m), _p(im), _p(pm), cap, 0)
        rc = self.L.vo_seq_run_multi(self.h, ptrs, counts, n, int(mem), int(w), int(h), int(stride),
                                     C.c_size_t(frame_stride), _p(K), C.byref(params), resp, outs)
        if rc != 0:
            raise VoError(rc, self.L.vo_seq_last_error(self.h).decode())
        return res

    def run(self, frames, n_frames, w, h, stride, frame_stride, K, params, mem=MEM_DEVICE, outputs=None):
        """frames: integer address (device or pinned host) or a uint8 numpy array (pageable host).
        outputs: optional (matches (n-1, cap) DM_DTYPE, inlier_mask (n-1, cap) uint8, pose_mask (n-1, cap) uint8)
        host arrays filled with the per-pair tables (any of them may be None)."""
        if isinstance(frames, np.ndarray):
            assert frames.dtype == np.uint8 and frames.flags.c_contiguous
            ptr = frames.ctypes.data_as(C.c_void_p)
        else:
            ptr = C.c_void_p(int(frames))
        K = np.ascontiguousarray(K, np.float64)
        res = (PairResult * max(n_frames - 1, 1))()
        out = None
        if outputs is not None:
            m, im, pm = outputs
            cap = next(a.shape[1] for a in (m, im, pm) if a is not None)
            for a in (m, im, pm):
                assert a is None or (a.flags.c_contiguous and a.shape == (n_frames - 1, cap))
            out = C.byref(SeqOutputs(_p(m), _p(im), _p(pm), cap, 0))
        rc = self.L.vo_seq_run_ex(self.h, ptr, int(mem), int(n_frames), int(w), int(h), int(stride),
                                  C.c_size_t(frame_stride), _p(K), C.byref(params), res, out)
        if rc != 0:
            raise VoError(rc, self.L.vo_seq_last_error(self.h).decode())
        return res
