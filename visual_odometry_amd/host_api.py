"""ctypes binding of libvo_host.so's C entry points (visual_odometry_amd/host/vo_host_c.h): the C++
host mirror of the reference's VisualOdometry driver, FeatureTracker / removePoints and the KITTI
formats, for tests and the sequence tool.  Plumbing only: the driver logic is the C++ class.
Nothing here imports oracle/."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvo_host.so")
SYMBOLS = ["vo_host_vo_create", "vo_host_vo_destroy", "vo_host_vo_add_image", "vo_host_vo_state",
           "vo_host_vo_arrays", "vo_host_load_calibration", "vo_host_load_poses", "vo_host_save_poses",
           "vo_host_remove_points", "vo_host_create_match_matrix", "vo_host_ba_create", "vo_host_ba_create_ex", "vo_host_ba_desc_bytes", "vo_host_ba_destroy",
           "vo_host_ba_add_key_frame", "vo_host_ba_counts", "vo_host_ba_features", "vo_host_ba_pair",
           "vo_host_ba_match_matrix", "vo_host_ba_points", "vo_host_vo_run_concurrent", "vo_host_timing_enable",
           "vo_host_timing_reset", "vo_host_timing_get"]
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libvo_host.so is not built (make -C visual_odometry_amd/host)")
        _lib = C.CDLL(LIB_PATH)
        _lib.vo_host_vo_create.restype = C.c_void_p
        _lib.vo_host_vo_create.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_char_p, C.c_int]
        _lib.vo_host_vo_destroy.argtypes = [C.c_void_p]
        _lib.vo_host_vo_destroy.restype = None
        _lib.vo_host_vo_add_image.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_char_p, C.c_int]
        _lib.vo_host_vo_state.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        _lib.vo_host_vo_arrays.argtypes = [C.c_void_p] + [C.c_void_p] * 7
        _lib.vo_host_ba_create.restype = C.c_void_p
        _lib.vo_host_ba_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int]
        _lib.vo_host_ba_create_ex.restype = C.c_void_p
        _lib.vo_host_ba_create_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]
        _lib.vo_host_ba_desc_bytes.argtypes = [C.c_void_p]
        _lib.vo_host_ba_destroy.argtypes = [C.c_void_p]
        _lib.vo_host_ba_destroy.restype = None
        _lib.vo_host_ba_add_key_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                  C.c_char_p, C.c_int]
        _lib.vo_host_ba_counts.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        _lib.vo_host_ba_features.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _lib.vo_host_ba_pair.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_void_p]
        _lib.vo_host_ba_match_matrix.argtypes = [C.c_void_p, C.c_void_p]
        _lib.vo_host_ba_points.argtypes = [C.c_void_p] * 4
        _lib.vo_host_vo_run_concurrent.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p,
                                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                   C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
        _lib.vo_host_timing_enable.argtypes = [C.c_int]
        _lib.vo_host_timing_enable.restype = None
        _lib.vo_host_timing_reset.restype = None
        _lib.vo_host_timing_get.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class HostVisualOdometry:
    """VisualOdometry(K, min_tracked_points, ba=false) of the C++ host mirror (reference:
    src/visual_odometry/visual_odometry.cpp:9-68)."""

    def __init__(self, K, min_tracked_points=750, scale_seed=1234, tracker_channels=3):
        self.L = lib()
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(3, 3)
        err = C.create_string_buffer(512)
        self.h = self.L.vo_host_vo_create(_p(K), int(min_tracked_points), int(scale_seed), int(tracker_channels),
                                          err, 512)
        if not self.h:
            raise RuntimeError("VisualOdometry: " + err.value.decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.vo_host_vo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_image(self, gray):
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        h, w = gray.shape
        pose = np.zeros((3, 4))
        err = C.create_string_buffer(512)
        if self.L.vo_host_vo_add_image(self.h, _p(gray), w, h, _p(pose), err, 512) != 0:
            raise RuntimeError("addImage: " + err.value.decode())
        return pose

    def state(self):
        """dict with the last frame pair's points, mask, E, local_R, local_t, points_3d, scale."""
        full, tracking, n, nm = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        scale = C.c_float(1.0)
        self.L.vo_host_vo_state(self.h, C.byref(full), C.byref(tracking), C.byref(n), C.byref(nm), C.byref(scale))
        out = {"full": bool(full.value), "tracking": bool(tracking.value), "scale": scale.value}
        if not full.value:
            return out
        n, nm = n.value, nm.value
        p0 = np.zeros((max(n, 1), 2), np.float32)
        p1 = np.zeros((max(n, 1), 2), np.float32)
        mask = np.zeros(max(nm, 1), np.uint8)
        E, R, t = np.zeros((3, 3)), np.zeros((3, 3)), np.zeros(3)
        p3 = np.zeros((max(nm, 1), 3))
        self.L.vo_host_vo_arrays(self.h, _p(p0), _p(p1), _p(mask), _p(E), _p(R), _p(t), _p(p3))
        out.update(prev_points=p0[:n], points=p1[:n], mask=mask[:nm], E=E, local_R=R, local_t=t, points_3d=p3[:nm])
        return out


def run_concurrent(K, sequences, min_tracked_points=750, scale_seed=1234, tracker_channels=3, warmup=0,
                   want_poses=False, lockstep=False):
    """S VisualOdometry drivers side by side, one host thread each (vo_host_vo_run_concurrent).  sequences: list
    of (n, h, w) uint8 arrays of one frame size.  lockstep = G: the drivers form G groups whose device calls go down as
    one batch per frame step (host_batch.h), same results.  Returns (wall seconds of the timed part, frames in it, poses or
    None)."""
    L = lib()
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(3, 3)
    seqs = [np.ascontiguousarray(s, dtype=np.uint8) for s in sequences]
    n = len(seqs)
    h, w = seqs[0].shape[1:]
    assert all(s.ndim == 3 and s.shape[1:] == (h, w) for s in seqs)
    ptrs = (C.c_void_p * n)(*[s.ctypes.data for s in seqs])
    counts = (C.c_int * n)(*[len(s) for s in seqs])
    poses = [np.zeros((len(s), 12)) for s in seqs] if want_poses else None
    pp = (C.c_void_p * n)(*[p.ctypes.data for p in poses]) if want_poses else None
    wall, timed = C.c_double(0), C.c_long(0)
    err = C.create_string_buffer(512)
    rc = L.vo_host_vo_run_concurrent(_p(K), int(min_tracked_points), int(scale_seed), int(tracker_channels), ptrs, counts,
                                     n, w, h, int(warmup), int(lockstep), pp, C.byref(wall), C.byref(timed), err, 512)
    if rc != 0:
        raise RuntimeError("run_concurrent: " + err.value.decode())
    return wall.value, timed.value, poses


def timing_enable(on=True):
    lib().vo_host_timing_enable(int(bool(on)))
    lib().vo_host_timing_reset()


def timing_get():
    """{stage name: (calls, microseconds per call)} of the per-frame driver since timing_enable()."""
    L = lib()
    out = {}
    name, calls, us = C.c_char_p(), C.c_long(0), C.c_double(0)
    n = L.vo_host_timing_get(0, C.byref(name), C.byref(calls), C.byref(us))
    for i in range(n):
        L.vo_host_timing_get(i, C.byref(name), C.byref(calls), C.byref(us))
        if calls.value:
            out[name.value.decode()] = (calls.value, us.value)
    return out


def load_calibration(path, line_number=0):
    K = np.zeros((3, 3))
    err = C.create_string_buffer(512)
    if lib().vo_host_load_calibration(path.encode(), int(line_number), _p(K), err, 512) != 0:
        raise RuntimeError(err.value.decode())
    return K


def load_poses(path):
    n = C.c_int(0)
    lib().vo_host_load_poses(path.encode(), None, 0, C.byref(n))
    poses = np.zeros((max(n.value, 1), 3, 4))
    lib().vo_host_load_poses(path.encode(), _p(poses), n.value, C.byref(n))
    return poses[: n.value]


def save_poses(path, poses):
    poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 3, 4)
    if lib().vo_host_save_poses(path.encode(), _p(poses), len(poses)) != 0:
        raise RuntimeError("cannot write " + path)


def remove_points(p0, p1, status, pts3d=None, mask=None):
    p0 = np.array(p0, dtype=np.float32, copy=True).reshape(-1, 2)
    p1 = np.array(p1, dtype=np.float32, copy=True).reshape(-1, 2)
    st = np.array(status, dtype=np.uint8, copy=True)
    p3 = np.array(pts3d if pts3d is not None else np.zeros((0, 3)), dtype=np.float64, copy=True).reshape(-1, 3)
    mk = np.array(mask if mask is not None else np.zeros(0), dtype=np.uint8, copy=True).reshape(-1)
    n_out, n3, nm, ret = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    lib().vo_host_remove_points(_p(p0), _p(p1), len(p0), _p(st), _p(p3), len(p3), _p(mk), len(mk), C.byref(n_out),
                                C.byref(n3), C.byref(nm), C.byref(ret))
    return p0[: n_out.value].copy(), p1[: n_out.value].copy(), st, p3[: n3.value].copy(), mk[: nm.value].copy(), ret.value


# ---- KITTI odometry error metric (published definition; the reference's call to it is commented
# out at src/kitti/kitti.cpp:65-70 and its devkit copy is third-party code that is not reproduced) --
KITTI_LENGTHS = (100, 200, 300, 400, 500, 600, 700, 800)


def kitti_sequence_errors(gt, res, step=10, lengths=KITTI_LENGTHS):
    """Mean translational (fraction) and rotational (rad/m) error over all sub-sequences of the
    given lengths, starting every `step` frames.  gt/res: N x 3 x 4 camera-to-world poses."""
    def to4(P):
        M = np.tile(np.eye(4), (len(P), 1, 1))
        M[:, :3, :] = P
        return M

    gt, res = to4(np.asarray(gt)), to4(np.asarray(res))
    n = min(len(gt), len(res))
    dist = np.concatenate([[0.0], np.cumsum(np.linalg.norm(np.diff(gt[:n, :3, 3], axis=0), axis=1))])
    t_err, r_err = [], []
    for first in range(0, n, step):
        for L in lengths:
            last = np.searchsorted(dist, dist[first] + L, side="right")  # first frame strictly beyond the length
            if last >= n:
                continue
            dgt = np.linalg.inv(gt[first]) @ gt[last]
            dres = np.linalg.inv(res[first]) @ res[last]
            e = np.linalg.inv(dres) @ dgt
            d = max(min(0.5 * (np.trace(e[:3, :3]) - 1.0), 1.0), -1.0)
            r_err.append(np.arccos(d) / L)
            t_err.append(np.linalg.norm(e[:3, 3]) / L)
    if not t_err:
        return None
    return float(np.mean(t_err)), float(np.mean(r_err)), len(t_err)


# ---- feature tracks and the bundle-adjustment front end ------------------------------------------------
def create_match_matrix(pairs, pose_count):
    """createMatchMatrix of the C++ mirror.  pairs: list of (src_img_idx, dst_img_idx, queryIdx[], trainIdx[],
    inliers_mask[]).  Returns the rows x pose_count int matrix (-1 = no keypoint)."""
    src = np.array([p[0] for p in pairs], np.int32)
    dst = np.array([p[1] for p in pairs], np.int32)
    off = np.concatenate([[0], np.cumsum([len(p[2]) for p in pairs])]).astype(np.int32)
    q = np.concatenate([np.asarray(p[2], np.int32) for p in pairs] + [np.zeros(0, np.int32)])
    t = np.concatenate([np.asarray(p[3], np.int32) for p in pairs] + [np.zeros(0, np.int32)])
    inl = np.concatenate([np.asarray(p[4], np.uint8) for p in pairs] + [np.zeros(0, np.uint8)])
    cap = max(len(q), 1)
    out = np.full((cap, pose_count), -7, np.int32)
    n = C.c_int(0)
    rc = lib().vo_host_create_match_matrix(_p(src), _p(dst), _p(off), len(pairs), _p(q), _p(t), _p(inl), int(pose_count),
                                           _p(out), cap, C.byref(n))
    if rc != 0:
        raise RuntimeError("createMatchMatrix failed (image index outside pose_count?)")
    return out[: n.value].copy()


class HostBundleAdjustment:
    """BundleAdjustment front end of the C++ host mirror (reference: src/sfm/bundle_adjustment.cpp:16-130):
    init + addKeyFrame; match matrix and setPBAPoints() outputs for an optimiser."""

    KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                         ("octave", "<i4"), ("class_id", "<i4")])
    DM_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])

    def __init__(self, K, max_frames=4, nfeatures=2000, descriptor="akaze"):
        """descriptor: "akaze" (the reference's addKeyFrame) or "orb" (nfeatures applies)."""
        self.L = lib()
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(3, 3)
        err = C.create_string_buffer(512)
        self.h = self.L.vo_host_ba_create_ex(_p(K), int(max_frames), 0 if descriptor == "akaze" else 1,
                                             int(nfeatures), err, 512)
        if not self.h:
            raise RuntimeError("BundleAdjustment: " + err.value.decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.vo_host_ba_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_key_frame(self, gray, pose_R, pose_t):
        gray = np.ascontiguousarray(gray, np.uint8)
        R = np.ascontiguousarray(pose_R, np.float64).reshape(3, 3)
        t = np.ascontiguousarray(pose_t, np.float64).reshape(3)
        err = C.create_string_buffer(512)
        if self.L.vo_host_ba_add_key_frame(self.h, _p(gray), gray.shape[1], gray.shape[0], _p(R), _p(t), err, 512) != 0:
            raise RuntimeError("addKeyFrame: " + err.value.decode())

    def counts(self):
        v = [C.c_int(0) for _ in range(4)]
        self.L.vo_host_ba_counts(self.h, *[C.byref(x) for x in v])
        return tuple(x.value for x in v)  # frames, pairs, tracks, points

    def features(self, frame, cap=32768):
        kp = np.zeros(cap, self.KP_DTYPE)
        desc = np.zeros((cap, self.L.vo_host_ba_desc_bytes(self.h)), np.uint8)
        n = C.c_int(0)
        assert self.L.vo_host_ba_features(self.h, int(frame), _p(kp), _p(desc), cap, C.byref(n)) == 0
        return kp[: n.value].copy(), desc[: n.value].copy()

    def pair(self, p, cap=32768):
        m = np.zeros(cap, self.DM_DTYPE)
        inl = np.zeros(cap, np.uint8)
        src, dst, n = C.c_int(0), C.c_int(0), C.c_int(0)
        assert self.L.vo_host_ba_pair(self.h, int(p), C.byref(src), C.byref(dst), _p(m), _p(inl), cap, C.byref(n)) == 0
        return src.value, dst.value, m[: n.value].copy(), inl[: n.value].copy()

    def match_matrix(self):
        nf, _, nr, _ = self.counts()
        out = np.zeros((max(nr, 1), nf), np.int32)
        self.L.vo_host_ba_match_matrix(self.h, _p(out))
        return out[:nr]

    def points(self):
        nf, _, _, npts = self.counts()
        P = np.zeros((max(npts, 1), 3))
        vis = np.zeros((max(npts, 1), nf), np.uint8)
        img = np.zeros((max(npts, 1), nf, 2), np.float32)
        self.L.vo_host_ba_points(self.h, _p(P), _p(vis), _p(img))
        return P[:npts], vis[:npts], img[:npts]
