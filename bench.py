#!/usr/bin/env python3
"""bench.py — frame-pairs/s of the per-frame-pair hot path on N MI355X (BASELINE.json metric).

N = 1 (BASELINE.json configs[3]): one seq-00-shaped sequence, 4541 synthetic KITTI-shaped 1241x376 gray
frames resident in HBM, through vo_seq_run_ex(): ORB nfeatures=2000 detect+describe on every frame,
brute-force Hamming 2-NN + ratio 0.7 + 200 px gate between consecutive frames, findEssentialMat(RANSAC,
0.999, 1.0, seed (uint64)-1), recoverPose(200); chunks of `--chunk` pairs on `--streams` HIP streams; the
per-pair results, filtered matches, inlier masks and pose masks come back to the host inside the timed
region.  A step = one pass over the whole sequence (4540 frame pairs).

N > 1 (configs[4]): the 22 KITTI-shaped sequences (lengths: reference src/kitti/readme.txt:125-135 for
00-10, SURVEY.md 8d for 11-21) are assigned to the ranks by longest-processing-time-first
(visual_odometry_amd/shard.py), every rank runs its sequences the same way, no data-path collective; the
trajectories are gathered over RCCL after the timed region.  Total work is fixed: strong scaling.
`python bench.py --gpus N` starts the N ranks itself (one process per GPU) unless a launcher
(torch.distributed.run) already did.

Frames: one `--base-frames`-frame sequence is rendered per rank (visual_odometry_amd/synth.py) and the long
sequences are cut from it back and forth (0..B-1, B-2..1, 0..), so every consecutive pair is a pair of
neighbouring rendered frames; every frame of the long sequence is its own 466 616 bytes in HBM.

Prints ONE JSON line on rank 0.  `roofline` is the matcher kernel — the Hamming distances as exact int8 dot
products on the matrix cores — against the dense int8 MFMA peak; `cpu_baseline` is the optimised
CPU port (cpu_baseline/) on the host cores on a bounded sample.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Four chunks in flight need a hardware queue each; HIP gives a process four by default (GPU_MAX_HW_QUEUES), one
# of which carries the default stream, so with the default the fourth compute stream shares a queue and the
# step is 18 % slower than with three streams, with eight it is 5 % faster (sweep in DESIGN.md, section 6).
# The HIP runtime reads the variable when it initialises, so it is set here, before anything touches the GPU
# (child ranks inherit it).  --hw-queues 0 leaves the environment as it is.
def _hw_queues_from_argv(default=8):
    for i, a in enumerate(sys.argv):
        if a == "--hw-queues" and i + 1 < len(sys.argv):
            return int(sys.argv[i + 1])
        if a.startswith("--hw-queues="):
            return int(a.split("=", 1)[1])
    return default


_HWQ = _hw_queues_from_argv()
if _HWQ > 0:
    os.environ["GPU_MAX_HW_QUEUES"] = str(min(_HWQ, 32))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

# Matcher on the matrix cores: one v_mfma_i32_32x32x32_i8 (32 train rows x 32 queries x 32 bit positions =
# 32768 multiply-adds) issues every 32 cycles on each of the 1024 SIMDs (tools/micro/mfma_rate.hip; the same
# cycles as the bf16 32x32x16 form, so twice its rate: MI355X_MICROARCH.md's ~2.5 PFLOP/s dense bf16 -> ~5 POP/s).
# A descriptor pair is 256 multiply-adds = 512 operations.
SIMDS, CLOCK_GHZ = 1024, 2.4
INT8_PEAK_TOPS = SIMDS * CLOCK_GHZ * 1e9 * (32 * 32 * 32 / 32.0) * 2 / 1e12  # 5033
OPS_PER_DESC_PAIR = 512
# What the instruction mix allows (tools/micro/mfma_valu.hip, profiles/r2_mfma_valu.txt): beside back-to-back
# MFMAs a SIMD spends ~17 cycles per MFMA plus ~2.8 per vector instruction past the fifth; the kernel needs
# ~10 vector instructions per MFMA (6 of them the three 16-bit min / max per candidate).
MIX_CYCLES_PER_MFMA = 17 + 2.8 * 10


def mfma_ceiling_tpairs():
    return INT8_PEAK_TOPS / OPS_PER_DESC_PAIR


def scan_bytes(nq, nt):
    return nq * nt * 32 + nq * 32 + nq * 16


def compulsory_bytes(nq, nt):
    return (nq + nt) * 32 + nq * 16


# ---------------------------------------------------------------------------------------------------
def self_launch(args):
    """--gpus N without a launcher: start the N ranks (fresh processes, before anything touched the GPU)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        for p in procs:
            rc = max(rc, abs(p.wait()))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def cpu_port():
    """The optimised CPU port (cpu_baseline/libvo_cpu.so, x86-64-v3) bound through oracle/pyoracle.py's
    ctypes wrappers; the plain oracle build if the host CPU lacks AVX2 / BMI2 / POPCNT."""
    import importlib.util

    from oracle import pyoracle as o

    path = os.path.join(ROOT, "cpu_baseline", "libvo_cpu.so")
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        flags = ""
    if not os.path.exists(path) or not all(f in flags for f in (" avx2", " bmi2", " popcnt", " fma")):
        return o, "oracle/ (-O2, byte-wise popcount: cpu_baseline/ unavailable on this host)"
    spec = importlib.util.spec_from_file_location("pyoracle_cpu_baseline", o.__file__)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m._LIB_PATH = path
    return m, "cpu_baseline/ (-O3 x86-64-v3, 64-bit popcount matcher, early-reject FAST, vectorised blur)"


def cpu_baseline(frames, K, nfeatures, max_iters=1000, budget_s=2.0, reps=20):
    """The same path on the host cores.  Per stage: median and min of `reps` single-thread repetitions after
    3 warm-ups.  Throughput: frame pairs are independent on this path, so worker threads
    (ctypes releases the GIL) take pairs off a shared counter until the budget is spent; value = pairs
    finished / wall time with `cores` threads."""
    import threading

    o, what = cpu_port()
    po = o.orb_params(nfeatures)

    def timed(fn):
        for _ in range(3):
            fn()
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t)
        return {"median_ms": float(np.median(ts)) * 1e3, "min_ms": float(np.min(ts)) * 1e3}

    k0, d0 = o.orb_detect_compute(frames[0], po)
    k1, d1 = o.orb_detect_compute(frames[1], po)
    knn, nv = o.knn2_hamming(d0, d1)
    good = o.match_filter(knn, nv, k0, k1, 0.7, 200.0)
    p0 = np.stack([k0["x"][good["queryIdx"]], k0["y"][good["queryIdx"]]], 1)
    p1 = np.stack([k1["x"][good["trainIdx"]], k1["y"][good["trainIdx"]]], 1)
    rc, E, mask, _ = o.find_essential(p0, p1, K, o.RANSAC, max_iters=max_iters)
    stages = {
        "orb_detect_compute_per_frame": timed(lambda: o.orb_detect_compute(frames[0], po)),
        "knn2_hamming_per_pair": timed(lambda: o.knn2_hamming(d0, d1)),
        "match_filter_per_pair": timed(lambda: o.match_filter(knn, nv, k0, k1, 0.7, 200.0)),
        "find_essential_ransac_per_pair": timed(lambda: o.find_essential(p0, p1, K, o.RANSAC, max_iters=max_iters)),
        "recover_pose_per_pair": timed(lambda: o.recover_pose(E, p0, p1, K, 200.0, mask)) if rc == 0 else None,
    }

    cores = max(1, min(16, os.cpu_count() or 1, len(frames) - 1))
    from visual_odometry_amd import synth

    vmax = 100000  # virtual sequence cut back and forth from the rendered frames, as on the GPU side
    vidx = synth.pingpong_indices(len(frames), vmax)
    t0 = time.perf_counter()
    lock = threading.Lock()
    state = {"next": 0, "done": 0, "busy_s": 0.0}
    feats = {}
    ready = {}
    used = {}

    def event(v):
        with lock:
            if v not in ready:
                ready[v] = threading.Event()
            return ready[v]

    def describe(v):
        feats[v] = o.orb_detect_compute(frames[vidx[v]], po)
        event(v).set()

    def one_pair(i):
        # every frame is described once: the worker of pair i makes frame i + 1 (pair 0 also frame 0)
        # and takes frame i from the worker of pair i - 1, which was handed out earlier
        if i == 0:
            describe(0)
        describe(i + 1)
        event(i).wait()
        (a0, b0), (a1, b1) = feats[i], feats[i + 1]
        kn, v = o.knn2_hamming(b0, b1)
        g = o.match_filter(kn, v, a0, a1, 0.7, 200.0)
        q0 = np.stack([a0["x"][g["queryIdx"]], a0["y"][g["queryIdx"]]], 1)
        q1 = np.stack([a1["x"][g["trainIdx"]], a1["y"][g["trainIdx"]]], 1)
        r, Em, m, _ = o.find_essential(q0, q1, K, o.RANSAC, max_iters=max_iters)
        if r == 0:
            o.recover_pose(Em, q0, q1, K, 200.0, m)
        with lock:  # a frame serves the pair before it and the pair after it
            for v in (i, i + 1):
                used[v] = used.get(v, 0) + 1
                if used[v] == (1 if v == 0 else 2):
                    feats.pop(v, None)
                    ready.pop(v, None)
                    used.pop(v, None)

    def worker():
        while True:
            with lock:
                i = state["next"]
                if i >= vmax - 1 or time.perf_counter() - t0 > budget_s:
                    return
                state["next"] = i + 1
            t1 = time.perf_counter()
            one_pair(i)
            with lock:
                state["done"] += 1
                state["busy_s"] += time.perf_counter() - t1

    threads = [threading.Thread(target=worker) for _ in range(cores)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    pairs = state["done"]
    return {"value": pairs / dt, "unit": "frame-pairs/s", "cores": cores, "kind": "port", "build": what,
            "per_thread_value": pairs / state["busy_s"] if state["busy_s"] > 0 else None,
            "stages_single_thread": stages, "stage_reps": reps,
            "sample": "%d consecutive frame pairs of the same sequence (ORB once per frame) on %d threads for %.1f s; "
                      "per-stage medians of %d single-thread repetitions" % (pairs, cores, dt, reps)}


# ---------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=0,
                    help="chunks in flight (HIP streams); 0 = 4 when the process has more than four hardware queues, else 3")
    ap.add_argument("--no-default-queues", action="store_true",
                    help="skip the child run that repeats the step with HIP's default number of hardware queues")
    ap.add_argument("--hw-queues", type=int, default=8,
                    help="GPU_MAX_HW_QUEUES for this process (read at the top of the file, before HIP starts); 0: leave it")
    ap.add_argument("--chunk", type=int, default=48, help="frame pairs per chunk")
    ap.add_argument("--frames", type=int, default=4541, help="frames of the N = 1 sequence (seq 00: 4541)")
    ap.add_argument("--base-frames", type=int, default=289, help="rendered frames the long sequences are cut from")
    ap.add_argument("--nfeatures", type=int, default=2000)
    ap.add_argument("--max-iters", type=int, default=1000,
                    help="RANSAC hypothesis cap (cv: 1000; BASELINE configs[2] stress: 4096 with --nfeatures 8192)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-8k", action="store_true", help="skip the 8192 x 8192 matcher side measurement")
    ap.add_argument("--no-pcie", action="store_true", help="skip the frames-in-pinned-host-memory side measurement")
    ap.add_argument("--no-configs4", action="store_true", help="N = 1: skip the 22-sequence configs[4] side run")
    ap.add_argument("--no-single-pair", action="store_true", help="skip the configs[1] single-pair latency measurement")
    ap.add_argument("--no-lk", action="store_true", help="skip the default (FAST -> LK -> LMedS) pipeline measurement")
    ap.add_argument("--lk-sequences", type=int, default=32, help="independent sequences of the concurrent LK run")
    ap.add_argument("--lk-groups", type=int, default=4, help="lock-step groups the concurrent LK drivers form")
    ap.add_argument("--workload", choices=["auto", "configs4"], default="auto",
                    help="auto: configs[3] on one GPU, configs[4] on several; configs4: the 22 sequences also at N = 1")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: launch, sharding, barriers and the trajectory gather only (CPU plumbing test)")
    args = ap.parse_args()
    explicit_streams = args.streams > 0
    if args.streams <= 0:
        args.streams = 4 if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) > 4 else 3

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: launch with `python bench.py --gpus N` (it starts the "
                         "ranks itself) or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`\n"
                         % (args.gpus, world))
        sys.exit(2)

    from visual_odometry_amd import shard, synth

    # ---- the same step in a child process that keeps HIP's default of four hardware queues (three chunks in
    # flight): what a host application gets if it does not raise GPU_MAX_HW_QUEUES before its first HIP call.
    # Run first, while this process has not touched the GPU; the rendered frames are shared through a cache.
    default_queues = None
    # (only for the default configuration: a run with --streams given, as the profiling scripts do, is not it)
    if (world == 1 and _HWQ > 0 and not args.no_default_queues and not args.dry_run and args.workload == "auto"
            and not explicit_streams):
        import atexit
        import shutil
        import tempfile

        tmp_cache = None
        if not os.environ.get("VO_SYNTH_CACHE"):
            tmp_cache = tempfile.mkdtemp(prefix="vo_synth_")
            os.environ["VO_SYNTH_CACHE"] = tmp_cache
            atexit.register(shutil.rmtree, tmp_cache, True)
        try:
            env = dict(os.environ)
            env.pop("GPU_MAX_HW_QUEUES", None)
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--hw-queues", "0", "--steps", "8",
                                  "--warmup", "2", "--frames", str(args.frames), "--base-frames",
                                  str(args.base_frames), "--chunk", str(args.chunk), "--nfeatures",
                                  str(args.nfeatures), "--max-iters", str(args.max_iters), "--no-cpu-baseline",
                                  "--no-8k", "--no-pcie", "--no-default-queues", "--no-configs4", "--no-single-pair",
                                  "--no-lk"],
                                 env=env, capture_output=True, text=True, timeout=150)
            child = json.loads(out.stdout.strip().splitlines()[-1])
            default_queues = {"value": child["value"], "unit": "frame-pairs/s",
                              "streams": child["config"]["streams"], "hw_queues": child["config"]["hw_queues"],
                              "note": "same step in a child process with GPU_MAX_HW_QUEUES unset (HIP's default: 4)"}
        except Exception as e:  # a side measurement: never fails the line
            default_queues = {"value": None, "note": "child run failed: %r" % (e,)}

    # ---- which sequences this rank runs ---------------------------------------------------------------
    multi = world > 1 or args.workload == "configs4"
    if not multi:
        my_seqs = [(0, args.frames)]
        lengths = [args.frames]
        imbalance = 1.0
    else:
        lengths = list(shard.KITTI_LENGTHS)
        assign = shard.lpt_assign(lengths, world)
        my_seqs = [(i, lengths[i]) for i in assign[rank]]
        loads = [sum(lengths[i] - 1 for i in a) for a in assign]
        imbalance = max(loads) / (sum(loads) / world)
    total_pairs_per_step = sum(n - 1 for n in lengths)
    lmax = max([n for _, n in my_seqs] + [2])
    # N = 1 also runs configs[4] (the 22 sequences through one pipeline) beside its headline: the buffer then
    # holds the longest of them
    side_configs4 = world == 1 and not multi and not args.no_configs4 and not args.dry_run
    if side_configs4:
        lmax = max(lmax, max(shard.KITTI_LENGTHS))

    # ---- synthetic input: rendered in forked workers before anything touches the GPU -------------------
    base_n = max(2, min(args.base_frames, lmax))
    if args.dry_run:
        frames = None
        K = synth.KITTI_K.copy()
    else:
        frames, _, K = synth.make_sequence(rank, base_n, workers=max(1, min(8, 16 // max(world, 1))))
    idx = synth.pingpong_indices(base_n, lmax)

    import torch

    # VO_BENCH_REHEARSE=1: every rank on GPU 0 and gloo instead of RCCL, to rehearse the multi-rank path
    # (launch, LPT shards, barriers, max over ranks, trajectory gather, rank-0 report) on a one-GPU box.
    rehearse = os.environ.get("VO_BENCH_REHEARSE", "0") == "1" or args.dry_run
    if rehearse:
        local_rank = 0
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    cdev = torch.device("cpu") if rehearse else torch.device("cuda", local_rank)  # where the collectives' tensors live

    def barrier():
        if dist is not None:
            dist.barrier()
        if not args.dry_run:
            torch.cuda.synchronize()

    if args.dry_run:
        # plumbing only: identity poses stand in for the results
        barrier()
        t0 = time.perf_counter()
        barrier()
        elapsed = time.perf_counter() - t0
        local = {sid: shard.chain_poses([(1, None, None)] * (n - 1), n) for sid, n in my_seqs}
        all_traj = shard.gather_trajectories(local, dist, cdev) if dist is not None else local
        assert sorted(all_traj) == list(range(len(lengths))) and all(len(all_traj[i]) == lengths[i] for i in all_traj)
        if rank == 0:
            print(json.dumps({"metric": "frame-pairs/sec (1241x376, 2k ORB)", "value": None, "unit": "frame-pairs/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True,
                              "scaling": "strong", "lpt_imbalance": imbalance, "sequences": len(lengths),
                              "trajectories_gathered": len(all_traj), "pairs_per_step": total_pairs_per_step,
                              "timed_region_s": elapsed}))
        if dist is not None:
            dist.destroy_process_group()
        return

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from visual_odometry_amd import api

    _, h, w = frames.shape
    d_base = torch.from_numpy(frames).to(dev)
    d_seq = d_base.index_select(0, torch.from_numpy(idx).to(dev))  # (lmax, h, w): every frame its own bytes
    del d_base
    torch.cuda.synchronize()

    seq = api.VoSequence(local_rank, n_streams=args.streams, pairs_per_chunk=args.chunk)
    ctx = seq.ctx(0)
    params = api.pair_params(args.nfeatures, max_iters=args.max_iters)
    cap = min(ctx.limits.max_points, args.nfeatures + args.nfeatures // 8 + 64)  # matches per pair <= keypoints per frame
    out_m = np.zeros((lmax - 1, cap), api.DM_DTYPE)
    out_im = np.zeros((lmax - 1, cap), np.uint8)
    out_pm = np.zeros((lmax - 1, cap), np.uint8)

    def run_seq(ptr, n, mem=api.MEM_DEVICE):
        return seq.run(ptr, n, w, h, w, w * h, K, params, mem=mem,
                       outputs=(out_m[: n - 1], out_im[: n - 1], out_pm[: n - 1]))

    def step():
        if len(my_seqs) == 1:
            return [run_seq(d_seq.data_ptr(), my_seqs[0][1])]
        # a rank's sequences go through one pipeline: the chunks of the next sequence follow without a drain
        return seq.run_multi([(d_seq.data_ptr(), n) for _, n in my_seqs], w, h, w, w * h, K, params,
                             outputs=[(out_m[: n - 1], out_im[: n - 1], out_pm[: n - 1]) for _, n in my_seqs])

    for _ in range(args.warmup):
        res = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()  # synchronous on return: results, matches and masks are on the host
    busy = time.perf_counter() - t0
    barrier()
    elapsed = time.perf_counter() - t0
    seq_stats = seq.stats()
    rank_busy = [busy]
    if dist is not None:
        t = torch.tensor([elapsed, busy], device=cdev, dtype=torch.float64)
        ts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(ts, t)
        elapsed = max(float(x[0]) for x in ts)
        rank_busy = [float(x[1]) for x in ts]
    total_pairs = args.steps * total_pairs_per_step

    # ---- chain the relative poses into trajectories on the host, gather over RCCL ----------------------
    local = {}
    for (sid, n), r in zip(my_seqs, res):
        local[sid] = shard.chain_poses([(r[p].status, np.array(r[p].R), np.array(r[p].t)) for p in range(n - 1)], n)
    all_traj = local
    if dist is not None:
        all_traj = shard.gather_trajectories(local, dist, cdev)
        torch.cuda.synchronize()
    assert sorted(all_traj) == list(range(len(lengths)))

    r0 = res[0]
    n0 = my_seqs[0][1]
    ok = sum(1 for r, (_, n) in zip(res, my_seqs) for p in range(n - 1) if r[p].status == 0)
    nkp = [r0[0].n_kp0] + [r0[p].n_kp1 for p in range(n0 - 1)]
    # the downloaded tables are the real ones: inlier counts of the last sequence run agree with its results
    rl, nl = res[-1], my_seqs[-1][1]
    for p in (0, nl // 2, nl - 2):
        assert int(out_im[p, : rl[p].n_good].sum()) == rl[p].n_inliers and int(out_pm[p, : rl[p].n_good].sum()) == rl[p].n_pose

    line = None
    if rank == 0:
        # ---- roofline: matcher kernel, HIP events on the library's stream (untimed passes) ------------
        # one chunk at a time on context 0, so that the matcher launch is timed without other streams
        # sharing the chip (the rocprofv3 summary in profiles/ is taken with --streams 1 for the same reason)
        nc = min(args.chunk, n0 - 1)  # pairs per matcher launch
        ctx.prof_enable(True)
        for which in (0, 1, 2, 3):
            ctx.prof_get(which, True)
        for _ in range(5):
            rb = ctx.pairs_run_dev(d_seq.data_ptr(), nc + 1, w, h, w, w * h, K, params)
        knn_n, knn_ms = ctx.prof_get(0)
        fast_n, fast_ms = ctx.prof_get(1)
        solve_n, solve_ms = ctx.prof_get(2)
        feat_n, feat_ms = ctx.prof_get(3)
        ctx.prof_enable(False)
        bkp = [rb[0].n_kp0] + [rb[p].n_kp1 for p in range(nc)]
        knn_avg_s = knn_ms / max(knn_n, 1) * 1e-3
        desc_pairs = sum(bkp[p] * bkp[p + 1] for p in range(nc))
        ceiling = mfma_ceiling_tpairs()
        achieved_pairs = desc_pairs / knn_avg_s / 1e12 if knn_avg_s > 0 else 0.0
        achieved = achieved_pairs * OPS_PER_DESC_PAIR  # TOP/s
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path):  # HBM bytes per launch from the committed PMC passes (tools/gpu_profile.sh)
            try:
                kern = json.load(open(pmc_path))["kernels"]
                traffic = [v for k, v in kern.items() if k.startswith("k_knn2")][0]["hbm_bytes"]
            except Exception:
                traffic = None
        # every frame's descriptor table once (it is the query side of one pair and the train side of the next:
        # one read serves both) + the 16-byte result per query
        comp = sum(bkp[p] * 32 for p in range(nc + 1)) + sum(bkp[p] * 16 for p in range(nc))
        scan = sum(scan_bytes(bkp[p], bkp[p + 1]) for p in range(nc))
        roofline = {
            "kernel": "k_knn2_mfma", "bound": "mfma", "achieved": achieved, "peak": INT8_PEAK_TOPS,
            "unit": "TOP/s", "frac": achieved / INT8_PEAK_TOPS, "traffic": traffic,
            "peak_model": "dense int8 MFMA: v_mfma_i32_32x32x32_i8 (32768 multiply-adds) per 32 cycles on each of %d "
                          "SIMDs at %.1f GHz; a descriptor pair = 256 multiply-adds = %d operations, exact integer "
                          "arithmetic" % (SIMDS, CLOCK_GHZ, OPS_PER_DESC_PAIR),
            "achieved_Tpairs_s": achieved_pairs, "peak_Tpairs_s": ceiling,
            "issue_mix_bound": {"Tpairs_s": ceiling * 32.0 / MIX_CYCLES_PER_MFMA,
                                "frac_of_it": achieved_pairs / (ceiling * 32.0 / MIX_CYCLES_PER_MFMA),
                                "note": "what the selection costs: ~10 vector instructions per MFMA (three 16-bit "
                                        "min / max per candidate) make a SIMD spend ~%.0f cycles per MFMA instead of 32 "
                                        "(tools/micro/mfma_valu.hip)" % MIX_CYCLES_PER_MFMA},
            "avg_launch_us": knn_avg_s * 1e6, "pairs_per_launch": nc, "descriptor_pairs_per_launch": desc_pairs,
            "compulsory_bytes_per_launch": comp,
            "traffic_over_compulsory": traffic / comp if traffic else None,
            "hbm": {"achieved_GBs": traffic / knn_avg_s / 1e9 if traffic and knn_avg_s > 0 else None,
                    "peak_GBs": HBM_PEAK_GBS,
                    "frac": traffic / knn_avg_s / 1e9 / HBM_PEAK_GBS if traffic and knn_avg_s > 0 else None},
            "scan_model": {"note": "SURVEY.md 8d's O(N^2) model nq*nt*32 + nq*48 bytes per pair: not a bound (the "
                                   "rows are re-read from registers, not from HBM), kept for reference only",
                           "bytes_per_launch": scan, "GBs": scan / knn_avg_s / 1e9 if knn_avg_s > 0 else None},
        }
        # detector + descriptor stage (pyramid .. rBRIEF) against its algorithmic bytes (SURVEY.md 8d:
        # W*H + 4*P + K*60 per frame, P = pyramid pixels): HBM-bound in the limit, VALU / latency bound today
        pyr_px = sum(int(round(w / 1.2 ** l)) * int(round(h / 1.2 ** l)) for l in range(8))
        feat_bytes = sum(w * h + 4 * pyr_px + k * 60 for k in bkp)
        feat_s = feat_ms / max(feat_n, 1) * 1e-3
        detector = {"stage": "pyramid + FAST/NMS + selection + IC angle + blur + rBRIEF, %d frames per batch" % (nc + 1),
                    "bound": "hbm", "achieved": feat_bytes / feat_s / 1e9 if feat_s > 0 else 0.0, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": feat_bytes / feat_s / 1e9 / HBM_PEAK_GBS if feat_s > 0 else 0.0,
                    "algorithmic_bytes_per_batch": feat_bytes, "avg_batch_us": feat_s * 1e6}
        # the stage's largest kernel against what bounds IT: vector-instruction issue (one wave64 instruction
        # per SIMD every 4 cycles); instruction count from the committed SQ pass, duration live from HIP events
        try:
            with open(os.path.join(ROOT, "profiles", "sq_latest.json")) as fh:
                sqj = json.load(fh)
            fast_insts = float(sqj["kernels"]["k_fast_nms"]["SQ_INSTS_VALU"])
            fast_s = fast_ms / max(fast_n, 1) * 1e-3
            peak = 1024 * 2.4e9 / 4  # wave instructions per second over the chip's 1 024 SIMDs at 2.4 GHz
            detector["fast_nms"] = {"bound": "valu", "achieved": fast_insts / fast_s / 1e9 if fast_s > 0 else None,
                                    "peak": peak / 1e9, "unit": "G wave-instructions/s",
                                    "frac": fast_insts / fast_s / peak if fast_s > 0 else None,
                                    "vector_instructions_per_launch": fast_insts, "avg_launch_us": fast_s * 1e6,
                                    "source": "profiles/%s_sq.json" % sqj.get("tag", "?")}
        except (OSError, KeyError, ValueError):
            detector["fast_nms"] = None

        # ---- the same sequence with the frames in pinned host memory: every chunk crosses PCIe inside the
        # timed region (its copy overlaps the kernels of the other chunks).  Beside `value`, never as it.
        pcie = None
        if world == 1 and not args.no_pcie:
            pinned = torch.from_numpy(frames[idx[:n0]]).pin_memory()
            run_seq(pinned.data_ptr(), n0, api.MEM_HOST_PINNED)
            torch.cuda.synchronize()
            psteps = max(2, args.steps // 5)
            tp = time.perf_counter()
            for _ in range(psteps):
                run_seq(pinned.data_ptr(), n0, api.MEM_HOST_PINNED)
            tp = time.perf_counter() - tp
            pcie = {"value": psteps * (n0 - 1) / tp, "unit": "frame-pairs/s", "steps": psteps,
                    "note": "frames in pinned host memory: upload, kernels and the download of results, matches and "
                            "masks all inside the timed region (SURVEY.md 8d's end-to-end definition)"}
            del pinned

        # ---- 8k x 8k matcher (BASELINE configs[2] shape), same accounting --------------------------------
        matcher_8k = None
        if not args.no_8k:
            q, t = synth.matcher_descriptors(8192)
            for _ in range(2):
                ctx.knn2_hamming(q, t)
            ctx.prof_enable(True)
            ctx.prof_get(0, True)
            for _ in range(5):
                ctx.knn2_hamming(q, t)
            kn, kms = ctx.prof_get(0)
            ctx.prof_enable(False)
            avg = kms / max(kn, 1) * 1e-3
            matcher_8k = {"avg_launch_us": avg * 1e6, "achieved_Tpairs_s": 8192 * 8192 / avg / 1e12,
                          "frac_of_mfma_peak": 8192 * 8192 / avg / 1e12 / ceiling,
                          "compulsory_bytes": compulsory_bytes(8192, 8192)}

        # ---- side measurement of a row next to the path: key-frame AKAZE ----------------------
        side = None
        if not args.no_8k:
            ts = []
            for f in frames[:6]:
                t0 = time.perf_counter()
                ka, _ = ctx.akaze_detect_compute(f)
                ts.append(time.perf_counter() - t0)
            side = {"akaze_ms_per_frame": float(np.median(ts[1:])) * 1e3, "akaze_keypoints": int(len(ka)),
                    "note": "vo_akaze_detect_compute on one 1241x376 frame, host pointers in and out (key-frame "
                            "operation of the BA front end, not part of `value`)"}

        # ---- BASELINE configs[4] on this ONE GPU: the 22 KITTI-shaped sequences through one pipeline
        # (vo_seq_run_multi), every sequence a window of the resident buffer.  The base of the 1 -> 8 curve.
        configs4 = None
        if side_configs4:
            kl = list(shard.KITTI_LENGTHS)
            c4 = [(d_seq.data_ptr(), n) for n in kl]
            c4_out = [(out_m[: n - 1], out_im[: n - 1], out_pm[: n - 1]) for n in kl]
            seq.run_multi(c4, w, h, w, w * h, K, params, outputs=c4_out)
            torch.cuda.synchronize()
            c4_steps = max(2, args.steps // 5)
            tc = time.perf_counter()
            for _ in range(c4_steps):
                r4 = seq.run_multi(c4, w, h, w, w * h, K, params, outputs=c4_out)
            tc = time.perf_counter() - tc
            c4_pairs = sum(n - 1 for n in kl)
            c4_ok = sum(1 for r, n in zip(r4, kl) for p in range(n - 1) if r[p].status == 0)
            imb = {}
            for g in (1, 2, 4, 8):
                loads = [sum(kl[i] - 1 for i in a) for a in shard.lpt_assign(kl, g)]
                imb[str(g)] = max(loads) / (sum(loads) / g)
            configs4 = {"value": c4_steps * c4_pairs / tc, "unit": "frame-pairs/s", "steps": c4_steps,
                        "ms_per_step": tc / c4_steps * 1e3, "sequences": len(kl), "frames_per_step": sum(kl),
                        "pairs_per_step": c4_pairs, "pairs_with_pose": c4_ok, "lpt_imbalance": imb,
                        "note": "BASELINE configs[4] at N = 1: what `bench.py --gpus 1 --workload configs4` runs, beside "
                                "the configs[3] headline; lpt_imbalance = max rank load / mean for 1, 2, 4, 8 ranks"}

        # ---- BASELINE configs[1]: ONE frame pair (2 frames, 1241x376, 2k ORB) as latency, not throughput ------
        single = None
        if world == 1 and not args.no_single_pair:
            def med(fn, reps=60, skip=10):
                ts = []
                for i in range(reps):
                    t0 = time.perf_counter()
                    fn(i)
                    ts.append(time.perf_counter() - t0)
                return float(np.median(ts[skip:])) * 1e3

            nb = min(40, n0 - 1)
            ms_res = med(lambda i: ctx.pairs_run_dev(d_seq.data_ptr() + (i % nb) * w * h, 2, w, h, w, w * h, K, params))
            orbp = api.orb_params(args.nfeatures)

            def host_pair(i):
                f0, f1 = frames[idx[i % nb]], frames[idx[i % nb + 1]]
                k0, d0 = ctx.orb_detect_compute(f0, orbp)
                k1, d1 = ctx.orb_detect_compute(f1, orbp)
                knn, nv = ctx.knn2_hamming(d0, d1)
                good = ctx.match_filter(knn, nv, k0, k1, 0.7, 200.0)
                p0 = np.stack([k0["x"][good["queryIdx"]], k0["y"][good["queryIdx"]]], 1)
                p1 = np.stack([k1["x"][good["trainIdx"]], k1["y"][good["trainIdx"]]], 1)
                rc, E, mask = ctx.find_essential(p0, p1, K, max_iters=args.max_iters)
                if rc == 0:
                    ctx.recover_pose(E, p0, p1, K, 200.0, mask)

            ms_host = med(host_pair, reps=40, skip=8)
            ctx.prof_enable(True)
            for which in (0, 1, 2, 3):
                ctx.prof_get(which, True)
            for i in range(20):
                ctx.pairs_run_dev(d_seq.data_ptr() + (i % nb) * w * h, 2, w, h, w, w * h, K, params)
            kus = {}
            for name, which in (("features_2_frames", 3), ("fast_nms", 1), ("knn2", 0), ("solve5_rounds", 2)):
                cnt, ms = ctx.prof_get(which)
                kus[name] = ms / 20 * 1e3
            ctx.prof_enable(False)
            single = {"ms_resident": ms_res, "ms_host_pointers": ms_host, "kernel_us": kus,
                      "note": "configs[1]: 2 frames through vo_pairs_run_dev (frames in HBM, results on the host), median "
                              "of 50; ms_host_pointers: the reference's own call sequence through the host-pointer entry "
                              "points (detectAndCompute x2, knnMatch, filter, findEssentialMat, recoverPose: six "
                              "synchronous calls, numpy glue included); kernel_us: HIP events around the stage's launches"}

        # ---- the reference's DEFAULT pipeline (src/visual_odometry/visual_odometry.cpp:33-49: FAST when lost ->
        # pyramidal LK -> removePoints -> findEssentialMat(LMEDS) -> recoverPose -> chaining) through the C++
        # driver of the host mirror, one frame at a time; then S independent sequences side by side
        lk = None
        if world == 1 and not args.no_lk:
            from visual_odometry_amd import host_api

            nlk = min(len(frames), 240)
            wu = min(40, nlk // 3)
            host_api.timing_enable(True)
            wall1, timed1, _ = host_api.run_concurrent(K, [frames[:nlk]], warmup=wu)
            stages = {k: round(v[1], 1) for k, v in host_api.timing_get().items()}
            host_api.timing_enable(False)
            S = max(1, args.lk_sequences)
            G = max(1, min(args.lk_groups, S))
            wallT, timedT, _ = host_api.run_concurrent(K, [frames[:nlk]] * min(S, 8), warmup=wu)
            wallS, timedS, _ = host_api.run_concurrent(K, [frames[:nlk]] * S, warmup=wu, lockstep=G)
            lk = {"frames_per_s": timed1 / wall1, "per_stage_us": stages,
                  "concurrent": {"sequences": S, "lockstep_groups": G, "frames_per_s": timedS / wallS,
                                 "speedup_over_one_sequence": (timedS / wallS) / (timed1 / wall1),
                                 "threads_only": {"sequences": min(S, 8), "frames_per_s": timedT / wallT}},
                  "frames_per_sequence": nlk, "warmup_frames": wu,
                  "note": "VisualOdometry::addImage of the host mirror (C++), 1241x376 frames in host memory, 3-channel "
                          "LK sums as the reference's CPU build.  `concurrent`: S drivers on S host threads "
                          "(vo_host_vo_run_concurrent) in G lock-step groups: the drivers of a group meet at their two "
                          "device calls per frame, which go down as ONE vo_lk_track_batch / vo_essential_pose_batch "
                          "(bit-identical per driver: tests/test_gpu_driver_concurrent.py); `threads_only`: every "
                          "driver makes its own calls on its own streams"}

        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(frames, K, args.nfeatures, args.max_iters)

        if not multi:
            workload = ("configs[3]: one seq-00-shaped sequence of %d synthetic 1241x376 frames resident in HBM "
                        "(cut back and forth from %d rendered frames), ORB nfeatures=%d + BF-Hamming knn2/ratio/gate + "
                        "5-pt RANSAC + recoverPose per consecutive pair; chunks of %d pairs on %d HIP streams; per-pair "
                        "results, matches, inlier and pose masks downloaded inside the timed region"
                        % (n0, base_n, args.nfeatures, args.chunk, args.streams))
        else:
            workload = ("configs[4]: 22 KITTI-shaped sequences (%d frames, each cut back and forth from the rank's %d "
                        "rendered frames) LPT-sharded over %d GPUs, per sequence as configs[3]; trajectories gathered "
                        "over RCCL after the timed region" % (sum(lengths), base_n, world))
        line = {
            "metric": "frame-pairs/sec (1241x376, 2k ORB)",
            "value": total_pairs / elapsed,
            "unit": "frame-pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "u8 features/descriptors, i8 x i8 -> i32 matrix-core Hamming matching (exact), f64 geometry",
            "data": "synthetic (seeded planar-world KITTI-shaped frames rendered by visual_odometry_amd/synth.py)",
            "config": {"workload": workload, "sequences": len(lengths), "frames_per_step": sum(lengths),
                       "pairs_per_step": total_pairs_per_step, "nfeatures": args.nfeatures,
                       "ransac_max_iters": args.max_iters, "streams": args.streams, "pairs_per_chunk": args.chunk,
                       "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "rendered_base_frames": base_n,
                       "parallelism": "sequence shards x%d (LPT by frame count), no data-path collective" % world},
            "timed_region_s": elapsed,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "matcher_8k": matcher_8k,
            "configs4_one_gpu": configs4,
            "single_pair": single,
            "lk_pipeline": lk,
            "side": side,
            "default_queues": default_queues,
            "detector_roofline": detector,
            "pcie_inclusive": pcie,
            "lpt_imbalance": imbalance,
            "rank_busy_s": rank_busy,
            "trajectories_gathered": len(all_traj),
            "stage_us": {"fast_nms_kernel_per_batch": fast_ms / max(fast_n, 1) * 1e3 if fast_n else None,
                         "knn2_kernel_per_batch": knn_avg_s * 1e6,
                         "solve5_kernels_per_batch": solve_ms / 5 * 1e3,
                         "features_per_batch": feat_s * 1e6},
            "host_ms_last_sequence": {"chunks": seq_stats[0], "queueing": seq_stats[1], "waiting": seq_stats[2]},
            "pairs_with_pose": ok,
            "keypoints_per_frame": float(np.mean(nkp)),
        }
    if dist is not None:
        dist.barrier()
    if rank == 0:
        print(json.dumps(line))
    seq.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
