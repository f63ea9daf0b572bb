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
           "vo_host_remove_points"]
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
