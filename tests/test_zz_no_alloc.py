"""vo_knn2_hamming_static (the entry point that allocates nothing: its partial and merged keys live in a
static device array) against vo_knn2_hamming, bit for bit, through the C ABI: two calls in a row on the
same GPU, the second smaller than the first, so that words the first call left in the shared array
cannot leak into the second.  No counterpart in the reference; see include/vo_hip.h."""
import numpy as np
import pytest
import torch

from visual_odometry_amd import api, synth

pytestmark = pytest.mark.gpu


def _static(vo, q, t):
    dq = torch.from_numpy(q).to("cuda:0")
    dt = torch.from_numpy(t).to("cuda:0") if len(t) else torch.zeros(16, dtype=torch.uint8, device="cuda:0")
    out = torch.zeros(max(len(q), 1) * 2 * 16, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    vo.knn2_hamming_static(dq.data_ptr(), len(q), dt.data_ptr(), len(t), q.shape[1], out.data_ptr())
    vo.synchronize()
    return out.cpu().numpy().view(api.DM_DTYPE).reshape(-1, 2)[: len(q)]


def _same(a, b):
    for f in ("queryIdx", "trainIdx", "imgIdx", "distance"):
        assert np.array_equal(a[f], b[f]), f


def test_static_entry_point_equals_the_allocating_one_twice_in_a_row(vo, oracle):
    q, t = synth.matcher_descriptors(8192)
    want, _ = vo.knn2_hamming(q, t)
    _same(_static(vo, q, t), want)                      # 8192 x 8192: train range in 12 splits
    want2, _ = vo.knn2_hamming(q[:1000], t[:777])
    got2 = _static(vo, q[:1000], t[:777])
    _same(got2, want2)                                  # smaller, right after: nothing stale
    _same(got2, oracle.knn2_hamming(q[:1000], t[:777])[0])  # and the oracle's table itself, not only the sibling's
    want3, _ = vo.knn2_hamming(q[:70], t[:1])
    got3 = _static(vo, q[:70], t[:1])                   # fewer than two train rows: -1 / FLT_MAX
    _same(got3, want3)
    assert (got3["trainIdx"][:, 1] == -1).all()
    rng = np.random.default_rng(9)
    q64 = rng.integers(0, 256, (300, 64), dtype=np.uint8)
    t64 = rng.integers(0, 256, (2100, 64), dtype=np.uint8)
    want4, _ = vo.knn2_hamming(q64, t64)
    _same(_static(vo, q64, t64), want4)                 # 64-byte rows


def test_static_entry_point_rejects_what_it_cannot_hold(vo):
    d = torch.zeros(8193 * 32, dtype=torch.uint8, device="cuda:0")
    o = torch.zeros(8193 * 32, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(api.VoError) as e:
        vo.knn2_hamming_static(d.data_ptr(), 8193, d.data_ptr(), 100, 32, o.data_ptr())
    assert e.value.code == api.VO_ERR_CAPACITY
    with pytest.raises(api.VoError) as e:
        vo.knn2_hamming_static(d.data_ptr(), 100, d.data_ptr(), 100, 61, o.data_ptr())  # rows must be 32 or 64 bytes
    assert e.value.code == api.VO_ERR_INVALID
