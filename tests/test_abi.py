"""CPU tests of the drop-in boundary: libvo_hip.so loads, exports every symbol include/vo_hip.h
declares, shares its struct layouts with the Python binding, and fails loudly (no CPU fallback)
when no HIP device is present.  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from visual_odometry_amd import _lib, api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "vo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vo_[a-z0-9_]+)\s*\(", src)))


def test_header_and_symbol_list_agree():
    assert _declared_functions() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    for name in _declared_functions():
        assert hasattr(lib, name), name


def test_struct_layouts():
    assert api.KP_DTYPE.itemsize == 28 and api.DM_DTYPE.itemsize == 16  # cv::KeyPoint, cv::DMatch
    assert C.sizeof(_lib.OrbParams) == 36
    assert C.sizeof(_lib.PairParams) == 36 + 4 + 4 + 4 + 8 * 3 + 8 + 8  # with alignment padding
    assert C.sizeof(_lib.PairResult) == 24 + 8 * 21
    assert C.sizeof(_lib.LkParams) == 40  # vo_lk_params: 4 ints, 2 doubles, 2 ints
    lk = api.lk_params()
    assert (lk.win_w, lk.win_h, lk.max_level, lk.max_count, lk.channels) == (21, 21, 3, 30, 3)
    assert lk.epsilon == 0.01 and lk.min_eig_threshold == 1e-4
    p = api.orb_params(2000)
    assert (p.nfeatures, p.nlevels, p.edge_threshold, p.patch_size, p.fast_threshold) == (2000, 8, 31, 31, 20)
    assert abs(p.scale_factor - 1.2) < 1e-6
    pp = api.pair_params(2000)
    assert pp.seed == 2**64 - 1 and pp.max_iters == 1000 and abs(pp.ratio - 0.7) < 1e-6


def test_host_only_entry_point():
    # vo_projection_matrix is twelve flops on the host: usable without a device
    K = np.array([[700.0, 0, 600], [0, 700, 180], [0, 0, 1]])
    pose = np.hstack([np.eye(3), np.array([[1.0], [2.0], [3.0]])])
    P = np.zeros((3, 4))
    _lib.lib().vo_projection_matrix(K.ctypes.data_as(C.c_void_p), pose.ctypes.data_as(C.c_void_p),
                                    P.ctypes.data_as(C.c_void_p))
    assert np.allclose(P, K @ np.hstack([np.eye(3), -np.array([[1.0], [2.0], [3.0]])]))


def test_no_cpu_fallback_without_device():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.VoError):
        api.VoContext(0)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "visual_odometry_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in txt and "vo_oracle" not in txt and "libvo_oracle" not in txt, f
