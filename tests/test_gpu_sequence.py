"""GPU tests of the sequence runner (vo_seq_run, BASELINE.json configs[3]): a sequence cut into
chunks over several streams must give, pair for pair, exactly what one vo_pairs_run_dev call over
the same frames gives (which tests/test_gpu_pipeline.py pins to the oracle), for device, pageable
and strided host input; and the first pairs are checked against the oracle directly."""
import ctypes as C

import numpy as np
import pytest
import torch

from visual_odometry_amd import api, synth

pytestmark = pytest.mark.gpu

K = synth.KITTI_K
W, H = 640, 192
KS = K.copy()
KS[0, 2], KS[1, 2] = 320.0, 96.0


def as_tuple(r):
    return (r.n_kp0, r.n_kp1, r.n_good, r.n_inliers, r.n_pose, r.status, tuple(r.E), tuple(r.R), tuple(r.t))


@pytest.fixture(scope="module")
def seq_frames():
    fr, poses, _ = synth.make_sequence(3, 12, w=W, h=H, K=KS)
    return fr, poses


@pytest.fixture(scope="module")
def batch_results(seq_frames):
    fr, _ = seq_frames
    ctx = api.VoContext(0)
    d = torch.from_numpy(fr).to("cuda:0")
    torch.cuda.synchronize()
    n = fr.shape[0]
    res = ctx.pairs_run_dev(d.data_ptr(), n, W, H, W, W * H, KS, api.pair_params(500))
    out = [as_tuple(r) for r in res]
    ctx.close()
    return out


@pytest.mark.parametrize("streams,chunk", [(1, 11), (2, 4), (3, 3), (4, 1), (2, 32)])
def test_seq_device_equals_batch(seq_frames, batch_results, streams, chunk):
    fr, _ = seq_frames
    d = torch.from_numpy(fr).to("cuda:0")
    torch.cuda.synchronize()
    n = fr.shape[0]
    seq = api.VoSequence(0, n_streams=streams, pairs_per_chunk=chunk)
    for _ in range(2):  # second pass: contexts are reused with work history
        res = seq.run(d.data_ptr(), n, W, H, W, W * H, KS, api.pair_params(500))
        assert [as_tuple(r) for r in res] == batch_results
    seq.close()


def test_seq_host_pageable_and_strided(seq_frames, batch_results):
    fr, _ = seq_frames
    n = fr.shape[0]
    seq = api.VoSequence(0, n_streams=2, pairs_per_chunk=4)
    res = seq.run(np.ascontiguousarray(fr), n, W, H, W, W * H, KS, api.pair_params(500), mem=api.MEM_HOST)
    assert [as_tuple(r) for r in res] == batch_results
    # rows padded to 704 bytes, frames padded by 3 rows
    pad = np.full((n, H + 3, 704), 201, np.uint8)
    pad[:, :H, :W] = fr
    res = seq.run(pad, n, W, H, 704, 704 * (H + 3), KS, api.pair_params(500), mem=api.MEM_HOST)
    assert [as_tuple(r) for r in res] == batch_results
    seq.close()


@pytest.mark.parametrize("prefetch", [True, False])
def test_seq_host_pinned(seq_frames, batch_results, prefetch):
    """Frames in pinned host memory: uploaded on the chunk's own stream (default) or, with VO_OPT_SEQ_PREFETCH = 1,
    ahead of time on a copy stream; the same results either way."""
    fr, _ = seq_frames
    n = fr.shape[0]
    pinned = torch.from_numpy(fr).pin_memory()
    seq = api.VoSequence(0, n_streams=2, pairs_per_chunk=5)
    seq.ctx(0).set_option(api.OPT_SEQ_PREFETCH, int(prefetch))
    res = seq.run(pinned.data_ptr(), n, W, H, W, W * H, KS, api.pair_params(500), mem=api.MEM_HOST_PINNED)
    assert [as_tuple(r) for r in res] == batch_results
    seq.close()


def test_seq_against_oracle(seq_frames, oracle):
    fr, poses = seq_frames
    n = fr.shape[0]
    d = torch.from_numpy(fr).to("cuda:0")
    torch.cuda.synchronize()
    seq = api.VoSequence(0, n_streams=2, pairs_per_chunk=2)
    res = seq.run(d.data_ptr(), n, W, H, W, W * H, KS, api.pair_params(500))
    po = oracle.orb_params(500)
    feats = [oracle.orb_detect_compute(f, po) for f in fr[:5]]
    for p in range(4):
        (k0, d0), (k1, d1) = feats[p], feats[p + 1]
        knn, nv = oracle.knn2_hamming(d0, d1)
        good = oracle.match_filter(knn, nv, k0, k1, 0.7, 200.0)
        p0 = np.stack([k0["x"][good["queryIdx"]], k0["y"][good["queryIdx"]]], 1)
        p1 = np.stack([k1["x"][good["trainIdx"]], k1["y"][good["trainIdx"]]], 1)
        rc, E, mask, _ = oracle.find_essential(p0, p1, KS, oracle.RANSAC)
        r = res[p]
        assert (r.n_kp0, r.n_kp1, r.n_good, r.status) == (len(k0), len(k1), len(good), rc)
        if rc == 0:
            Eg = np.array(r.E).reshape(3, 3)
            assert min(np.abs(Eg - E).max(), np.abs(Eg + E).max()) < 1e-6
            assert r.n_inliers == int(mask.sum())
            ng, R, t, pm, _ = oracle.recover_pose(E, p0, p1, KS, 200.0, mask)
            assert r.n_pose == ng
            assert np.abs(np.array(r.R).reshape(3, 3) - R).max() < 1e-6  # north_star: 1e-6
            assert np.abs(np.array(r.t) - t).max() < 1e-6
    seq.close()


def test_seq_bad_arguments(seq_frames):
    fr, _ = seq_frames
    seq = api.VoSequence(0, n_streams=2, pairs_per_chunk=4)
    with pytest.raises(api.VoError):
        seq.run(np.ascontiguousarray(fr), 1, W, H, W, W * H, KS, api.pair_params(500), mem=api.MEM_HOST)
    with pytest.raises(api.VoError):
        seq.run(np.ascontiguousarray(fr), 4, W, H, W - 1, W * H, KS, api.pair_params(500), mem=api.MEM_HOST)
    with pytest.raises(api.VoError):
        seq.run(np.ascontiguousarray(fr), 4, W, H, W, W * H, KS, api.pair_params(500), mem=7)
    seq.close()


def _oracle_pair(oracle, f0, f1, nfeatures, Kc):
    po = oracle.orb_params(nfeatures)
    k0, d0 = oracle.orb_detect_compute(f0, po)
    k1, d1 = oracle.orb_detect_compute(f1, po)
    knn, nv = oracle.knn2_hamming(d0, d1)
    good = oracle.match_filter(knn, nv, k0, k1, 0.7, 200.0)
    p0 = np.stack([k0["x"][good["queryIdx"]], k0["y"][good["queryIdx"]]], 1)
    p1 = np.stack([k1["x"][good["trainIdx"]], k1["y"][good["trainIdx"]]], 1)
    rc, E, mask, _ = oracle.find_essential(p0, p1, Kc, oracle.RANSAC)
    ng, R, t, pm, _ = oracle.recover_pose(E, p0, p1, Kc, 200.0, mask)
    return dict(nk0=len(k0), nk1=len(k1), good=good, rc=rc, E=E, mask=mask, ng=ng, R=R, t=t, pm=pm)


def test_seq_outputs_equal_get_matches(seq_frames):
    """vo_seq_run_ex's match / mask tables are what vo_pairs_get_matches returns for the same frames."""
    fr, _ = seq_frames
    n = fr.shape[0]
    d = torch.from_numpy(fr).to("cuda:0")
    torch.cuda.synchronize()
    ctx = api.VoContext(0)
    ctx.pairs_run_dev(d.data_ptr(), n, W, H, W, W * H, KS, api.pair_params(500))
    want = [ctx.pairs_get_matches(p) for p in range(n - 1)]
    ctx.close()
    cap = 700
    m = np.zeros((n - 1, cap), api.DM_DTYPE)
    im = np.full((n - 1, cap), 9, np.uint8)
    pm = np.full((n - 1, cap), 9, np.uint8)
    seq = api.VoSequence(0, n_streams=3, pairs_per_chunk=4)
    res = seq.run(d.data_ptr(), n, W, H, W, W * H, KS, api.pair_params(500), outputs=(m, im, pm))
    for p in range(n - 1):
        g, a, b = want[p]
        k = res[p].n_good
        assert k == len(g) and np.array_equal(m[p, :k], g) and np.array_equal(im[p, :k], a) and np.array_equal(pm[p, :k], b)
        assert (im[p, k:] == 9).all()  # rows beyond n_good are left alone
    with pytest.raises(api.VoError) as e:  # a table too narrow for a pair's matches
        seq.run(d.data_ptr(), n, W, H, W, W * H, KS, api.pair_params(500), outputs=(m[:, :8].copy(), None, None))
    assert e.value.code == api.VO_ERR_CAPACITY
    seq.close()


def test_seq_full_length_sequence_configs3(oracle):
    """BASELINE.json configs[3] at its stated size: one seq-00-shaped sequence of 4541 frames (1241 x 376, 2k
    ORB) through vo_seq_run_ex in chunks of 48 pairs on 3 streams, the way bench.py runs it (the sequence is
    cut back and forth from 97 rendered frames).  Every pair must report a pose; the first chunk, the last
    chunk's last pair, a mid-sequence pair and a pair that straddles two chunks are checked against the
    oracle: matches bit-exact, masks bit-exact, E / R / t within north_star's 1e-6."""
    n, base = 4541, 97
    fr, _, Kc = synth.make_sequence(11, base, workers=8)
    idx = synth.pingpong_indices(base, n)
    d = torch.from_numpy(fr).to("cuda:0").index_select(0, torch.from_numpy(idx).to("cuda:0"))
    torch.cuda.synchronize()
    cap = 2314
    m = np.zeros((n - 1, cap), api.DM_DTYPE)
    im = np.zeros((n - 1, cap), np.uint8)
    pm = np.zeros((n - 1, cap), np.uint8)
    seq = api.VoSequence(0, n_streams=3, pairs_per_chunk=48)
    res = seq.run(d.data_ptr(), n, 1241, 376, 1241, 1241 * 376, Kc, api.pair_params(2000), outputs=(m, im, pm))
    assert seq.stats()[0] == 95
    seq.close()
    assert all(res[p].status == 0 for p in range(n - 1))
    assert all(res[p].n_kp0 >= 2000 and res[p].n_good > 100 and res[p].n_pose > 50 for p in range(n - 1))
    # pairs p and p' over the same two base frames in the same order give identical results wherever they sit
    period = 2 * (base - 1)
    for p in (0, 5, 47, 48, 100):
        a, b = res[p], res[p + 10 * period]
        assert (a.n_good, a.n_inliers, a.n_pose, tuple(a.E)) == (b.n_good, b.n_inliers, b.n_pose, tuple(b.E))
    for p in (0, 1, 47, 48, 2270, n - 2):  # 47 | 48: the frame two chunks share
        o = _oracle_pair(oracle, fr[idx[p]], fr[idx[p + 1]], 2000, Kc)
        r = res[p]
        k = r.n_good
        assert (r.n_kp0, r.n_kp1, k, r.status) == (o["nk0"], o["nk1"], len(o["good"]), o["rc"])
        assert np.array_equal(m[p, :k], o["good"])
        assert np.array_equal(im[p, :k], o["mask"]) and r.n_inliers == int(o["mask"].sum())
        assert np.array_equal(pm[p, :k], o["pm"]) and r.n_pose == o["ng"]
        Eg = np.array(r.E).reshape(3, 3)
        assert min(np.abs(Eg - o["E"]).max(), np.abs(Eg + o["E"]).max()) < 1e-6
        assert np.abs(np.array(r.R).reshape(3, 3) - o["R"]).max() < 1e-6 and np.abs(np.array(r.t) - o["t"]).max() < 1e-6


@pytest.mark.gpu
def test_seq_run_multi_equals_separate_runs():
    """vo_seq_run_multi pipelines the chunks of several sequences back to back: every result, match table and
    mask equals what one vo_seq_run_ex per sequence returns (device-resident and pinned-host frames)."""
    import torch

    from visual_odometry_amd import api, synth

    K = synth.KITTI_K.copy()
    K[0, 2], K[1, 2] = 320.0, 96.0
    frames, _, _ = synth.make_sequence(11, 23, w=640, h=192, K=K)
    lens = [23, 5, 2, 14]          # incl. a one-pair sequence and lengths that are not multiples of the chunk
    starts = [0, 3, 9, 7]
    d = torch.from_numpy(frames).cuda()
    pinned = torch.from_numpy(frames).pin_memory()
    n, h, w = frames.shape
    prm = api.pair_params(500)
    seq = api.VoSequence(0, n_streams=3, pairs_per_chunk=4)
    cap = 700
    for base, mem in ((d.data_ptr(), api.MEM_DEVICE), (pinned.data_ptr(), api.MEM_HOST_PINNED)):
        single = []
        for s0, ln in zip(starts, lens):
            o = tuple(np.zeros((ln - 1, cap), t) for t in (api.DM_DTYPE, np.uint8, np.uint8))
            r = seq.run(base + s0 * w * h, ln, w, h, w, w * h, K, prm, mem=mem, outputs=o)
            single.append((r, o))
        outs = [tuple(np.zeros((ln - 1, cap), t) for t in (api.DM_DTYPE, np.uint8, np.uint8)) for ln in lens]
        multi = seq.run_multi([(base + s0 * w * h, ln) for s0, ln in zip(starts, lens)], w, h, w, w * h, K, prm,
                              mem=mem, outputs=outs)
        for (r1, o1), r2, o2, ln in zip(single, multi, outs, lens):
            for p in range(ln - 1):
                for f in ("n_kp0", "n_kp1", "n_good", "n_inliers", "n_pose", "status"):
                    assert getattr(r1[p], f) == getattr(r2[p], f), (p, f)
                assert list(r1[p].E) == list(r2[p].E) and list(r1[p].R) == list(r2[p].R) and list(r1[p].t) == list(r2[p].t)
            for a, b in zip(o1, o2):
                assert np.array_equal(a, b)
    seq.close()


@pytest.mark.parametrize("ratio", [0.7, 0.97])
def test_pair_results_do_not_depend_on_the_chunk_size(seq_frames, ratio):
    """A pair's E, masks and pose must not depend on how many pairs share its launches (the
    five-point solver used to be chosen from hypotheses x pairs of a launch, and the two solvers are not
    bit-identical).  The same 48 pairs in one chunk of 48, in chunks of 5 and one pair at a time through
    vo_pairs_run_dev: identical results, bit for bit.  ratio 0.97 lets many wrong matches through, so the
    adaptive RANSAC runs its later, larger rounds (32 -> 128 -> 512 -> 328 hypotheses) as well."""
    fr, _ = seq_frames
    n = 49
    idx = synth.pingpong_indices(fr.shape[0], n)
    d = torch.from_numpy(fr[idx]).to("cuda:0")
    torch.cuda.synchronize()
    prm = api.pair_params(500, ratio=ratio)
    cap = 700
    runs = []
    for chunk in (48, 5):
        seq = api.VoSequence(0, n_streams=2, pairs_per_chunk=chunk)
        o = tuple(np.zeros((n - 1, cap), t) for t in (api.DM_DTYPE, np.uint8, np.uint8))
        res = seq.run(d.data_ptr(), n, W, H, W, W * H, KS, prm, outputs=o)
        runs.append(([as_tuple(r) for r in res], o))
        seq.close()
    (r48, o48), (r5, o5) = runs
    assert r48 == r5
    for a, b in zip(o48, o5):
        assert np.array_equal(a, b)
    assert all(r[5] == 0 for r in r48)
    ctx = api.VoContext(0)
    for p in (0, 7, 30, 47):
        one = ctx.pairs_run_dev(d.data_ptr() + p * W * H, 2, W, H, W, W * H, KS, prm)
        assert as_tuple(one[0]) == r48[p]
    ctx.close()


def test_seq_multi_configs4_full_shape(oracle):
    """BASELINE.json configs[4] at its stated shape on ONE GPU: the 22 KITTI sequence lengths
    (shard.KITTI_LENGTHS: 43 552 frames of 1241 x 376, 43 530 pairs) through vo_seq_run_multi in chunks of 48
    pairs on 3 streams.  Every sequence is a window of one shared ping-pong buffer (4 981 frames = the longest
    sequence), sequence i starting 3 i frames in, so memory stays at one buffer and the windows' chunk
    boundaries fall at different frames.  Checked: every pair has a pose; three sequences equal separate
    vo_seq_run_ex calls bit for bit; pair p of every sequence equals pair offset + p of the full-length
    sequence (results do not depend on which sequence or chunk a pair sits in); first / last pairs of
    sequences and pairs next to a sequence boundary match the oracle (matches and masks bit-exact, E / R / t
    within 1e-6)."""
    from visual_odometry_amd import shard

    lengths = list(shard.KITTI_LENGTHS)
    assert len(lengths) == 22 and sum(lengths) == 43552
    lmax, base = max(lengths), 97
    fr, _, Kc = synth.make_sequence(11, base, workers=8)
    idx = synth.pingpong_indices(base, lmax)
    d = torch.from_numpy(fr).to("cuda:0").index_select(0, torch.from_numpy(idx).to("cuda:0"))
    torch.cuda.synchronize()
    w, h = 1241, 376
    offs = [min(3 * i, lmax - n) for i, n in enumerate(lengths)]
    cap = 2314
    outs = [tuple(np.zeros((n - 1, cap), t) for t in (api.DM_DTYPE, np.uint8, np.uint8)) for n in lengths]
    prm = api.pair_params(2000)
    seq = api.VoSequence(0, n_streams=3, pairs_per_chunk=48)
    res = seq.run_multi([(d.data_ptr() + o * w * h, n) for o, n in zip(offs, lengths)], w, h, w, w * h, Kc, prm,
                        outputs=outs)
    assert seq.stats()[0] == sum((n - 1 + 47) // 48 for n in lengths)
    for r, n in zip(res, lengths):
        assert all(r[p].status == 0 and r[p].n_kp0 >= 2000 and r[p].n_good > 100 and r[p].n_pose > 50
                   for p in range(n - 1))
    # three sequences on their own (a short one, one that ends in a one-pair chunk, the first)
    for i in (4, 17, 0):
        n = lengths[i]
        o1 = tuple(np.zeros((n - 1, cap), t) for t in (api.DM_DTYPE, np.uint8, np.uint8))
        r1 = seq.run(d.data_ptr() + offs[i] * w * h, n, w, h, w, w * h, Kc, prm, outputs=o1)
        assert [as_tuple(r1[p]) for p in range(n - 1)] == [as_tuple(res[i][p]) for p in range(n - 1)]
        for a, b in zip(o1, outs[i]):
            assert np.array_equal(a, b)
    seq.close()
    # every sequence is a window of the longest one
    full = lengths.index(lmax)
    assert offs[full] == 0
    for i, n in enumerate(lengths):
        for p in list(range(0, n - 1, 97)) + [n - 2]:
            a, b = res[i][p], res[full][offs[i] + p]
            assert as_tuple(a) == as_tuple(b), (i, p)
            k = a.n_good
            for ta, tb in zip(outs[i], outs[full]):
                assert np.array_equal(ta[p, :k], tb[offs[i] + p, :k])
    # oracle: first and last pair of sequences, the pairs either side of a sequence boundary in the chunk stream
    for i, p in ((0, 0), (0, lengths[0] - 2), (1, 0), (4, lengths[4] - 2), (5, 0), (21, lengths[21] - 2), (19, 2500)):
        g0 = offs[i] + p
        o = _oracle_pair(oracle, fr[idx[g0]], fr[idx[g0 + 1]], 2000, Kc)
        r = res[i][p]
        k = r.n_good
        m, im, pm = outs[i]
        assert (r.n_kp0, r.n_kp1, k, r.status) == (o["nk0"], o["nk1"], len(o["good"]), o["rc"])
        assert np.array_equal(m[p, :k], o["good"])
        assert np.array_equal(im[p, :k], o["mask"]) and r.n_inliers == int(o["mask"].sum())
        assert np.array_equal(pm[p, :k], o["pm"]) and r.n_pose == o["ng"]
        Eg = np.array(r.E).reshape(3, 3)
        assert min(np.abs(Eg - o["E"]).max(), np.abs(Eg + o["E"]).max()) < 1e-6
        assert np.abs(np.array(r.R).reshape(3, 3) - o["R"]).max() < 1e-6 and np.abs(np.array(r.t) - o["t"]).max() < 1e-6


@pytest.mark.parametrize("mem", ["device", "pinned", "pageable"])
def test_seq_shared_frames_are_carried_over_also_through_the_host_redo(seq_frames, batch_results, mem):
    """Consecutive chunks share a frame; the runner gives every context a run of consecutive chunks and the
    context carries the shared frame's keypoints over instead of describing it again.  Chunks of 2 pairs on 2
    streams: most chunks start with a carried frame.  With a selection buffer of 64 entries every frame is
    flagged and redone on the host path, so the carried rows are the redone ones; with the default buffer none
    is.  Both must equal the one-batch results (which test_gpu_pipeline.py pins to the oracle), and the second
    pass over the same runner starts from contexts that hold stale chunks."""
    fr, _ = seq_frames
    n = fr.shape[0]
    keep = None
    if mem == "device":
        keep = torch.from_numpy(fr).to("cuda:0")
        torch.cuda.synchronize()
        src, m = keep.data_ptr(), api.MEM_DEVICE
    elif mem == "pinned":
        keep = torch.from_numpy(fr).pin_memory()
        src, m = keep.data_ptr(), api.MEM_HOST_PINNED
    else:
        src, m = fr, api.MEM_HOST
    for sel_cap in (None, 64):
        seq = api.VoSequence(0, n_streams=2, pairs_per_chunk=2)
        if sel_cap:
            for i in range(2):
                seq.ctx(i).set_option(api.OPT_SEL_CAP, sel_cap)
        for _ in range(2):
            res = seq.run(src, n, W, H, W, W * H, KS, api.pair_params(500), mem=m)
            assert [as_tuple(r) for r in res] == batch_results
            if sel_cap:
                assert all(seq.ctx(i).get_option(api.OPT_LAST_FLAGS) & 1 for i in range(2))
        seq.close()
