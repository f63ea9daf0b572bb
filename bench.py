#!/usr/bin/env python3
"""bench.py — frame-pairs/s of the per-frame-pair hot path on N MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1], batched): consecutive synthetic KITTI-shaped 1241x376 gray
frames, ORB nfeatures=2000 detect+describe on every frame, brute-force Hamming 2-NN + ratio 0.7 +
200 px gate between consecutive frames, findEssentialMat(RANSAC, 0.999, 1.0, seed (uint64)-1),
recoverPose(200).  A step = one pass of vo_seq_run() over `--frames` frames (frames-1 frame pairs)
already resident in HBM: the sequence runner cuts them into chunks of `--chunk` pairs and keeps
`--streams` chunks in flight on separate HIP streams (each chunk is one vo_pairs_run_dev batch).  With N > 1 every rank runs its own
synthetic sequence (sequences are the shard unit, no data-path collective; weak scaling) and the
trajectories are gathered with one all_gather over RCCL after the timed region.

Prints ONE JSON line on rank 0.  `roofline` is the matcher kernel under the O(N^2) scan model of
SURVEY.md §8(d) (nq*nt*32 + nq*32 + nq*16 bytes per pair), timed live with HIP events on the
library's stream; `cpu_baseline` is the CPU oracle (a port) on the host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def scan_bytes(nq, nt):
    return nq * nt * 32 + nq * 32 + nq * 16


def cpu_baseline(frames, K, nfeatures, max_iters=1000, budget_s=12.0):
    """The oracle's restatement of the same path on the host cores: frame pairs are independent on this
    path, so worker threads (ctypes releases the GIL inside the oracle) take pairs off a shared counter
    until the time budget is spent.  value = pairs finished / wall time with `cores` threads; the
    per-thread rate of the same run is reported next to it."""
    import threading

    from oracle import pyoracle as o

    po = o.orb_params(nfeatures)
    cores = max(1, min(16, os.cpu_count() or 1, len(frames) - 1))
    t0 = time.perf_counter()
    lock = threading.Lock()
    state = {"next": 0, "done": 0, "busy_s": 0.0}

    feats = [None] * len(frames)
    ready = [threading.Event() for _ in frames]

    def describe(i):
        feats[i] = o.orb_detect_compute(frames[i], po)
        ready[i].set()

    def one_pair(i):
        # every frame is described once: the worker of pair i makes frame i + 1 (pair 0 also frame 0)
        # and takes frame i from the worker of pair i - 1, which was handed out earlier
        if i == 0:
            describe(0)
        describe(i + 1)
        ready[i].wait()
        (k0, d0), (k1, d1) = feats[i], feats[i + 1]
        knn, nv = o.knn2_hamming(d0, d1)
        good = o.match_filter(knn, nv, k0, k1, 0.7, 200.0)
        p0 = np.stack([k0["x"][good["queryIdx"]], k0["y"][good["queryIdx"]]], 1)
        p1 = np.stack([k1["x"][good["trainIdx"]], k1["y"][good["trainIdx"]]], 1)
        rc, E, mask, _ = o.find_essential(p0, p1, K, o.RANSAC, max_iters=max_iters)
        if rc == 0:
            o.recover_pose(E, p0, p1, K, 200.0, mask)

    def worker():
        while True:
            with lock:
                i = state["next"]
                if i >= len(frames) - 1 or time.perf_counter() - t0 > budget_s:
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
    return {"value": pairs / dt, "unit": "frame-pairs/s", "cores": cores, "kind": "port",
            "per_thread_value": pairs / state["busy_s"] if state["busy_s"] > 0 else None,
            "sample": "%d frame pairs of the same batch (ORB once per frame), oracle/ on %d threads for %.1f s"
                      % (pairs, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=3, help="chunks in flight (HIP streams)")
    ap.add_argument("--chunk", type=int, default=48, help="frame pairs per chunk")
    ap.add_argument("--frames", type=int, default=0, help="frames per step (default 6 chunks: 6*chunk + 1)")
    ap.add_argument("--nfeatures", type=int, default=2000)
    ap.add_argument("--max-iters", type=int, default=1000,
                    help="RANSAC hypothesis cap (cv: 1000; BASELINE configs[2] stress: 4096 with --nfeatures 8192)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-8k", action="store_true", help="skip the 8192 x 8192 matcher side measurement")
    args = ap.parse_args()

    if args.frames <= 0:
        args.frames = 6 * args.chunk + 1  # two rounds of chunks per stream: the pipeline reaches steady state

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    # ---- synthetic input: each rank renders its own sequence (the shard unit); rendered in forked
    # workers before anything touches the GPU ------------------------------------------------------
    from visual_odometry_amd import synth

    frames, poses, K = synth.make_sequence(rank, args.frames, workers=max(1, min(8, 16 // max(world, 1))))

    import torch

    # VO_BENCH_REHEARSE=1: every rank on GPU 0 and gloo instead of RCCL, to rehearse the multi-rank path
    # (barriers, max over ranks, trajectory gather, rank-0 report) on a one-GPU box.  Not a measurement.
    rehearse = os.environ.get("VO_BENCH_REHEARSE", "0") == "1"
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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if rehearse else dev  # where the collectives' tensors live

    from visual_odometry_amd import api, shard

    n, h, w = frames.shape
    d_frames = torch.from_numpy(frames).to(dev)
    torch.cuda.synchronize()

    seq = api.VoSequence(local_rank, n_streams=args.streams, pairs_per_chunk=args.chunk)
    ctx = seq.ctx(0)
    params = api.pair_params(args.nfeatures, max_iters=args.max_iters)

    def step():
        return seq.run(d_frames.data_ptr(), n, w, h, w, w * h, K, params)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        res = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()  # synchronous on return: results are on the host
    barrier()
    elapsed = time.perf_counter() - t0
    seq_stats = seq.stats()
    if dist is not None:
        t = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    pairs_per_step = n - 1
    total_pairs = world * args.steps * pairs_per_step

    # ---- chain the relative poses into a trajectory on the host, gather over RCCL --------------
    traj = shard.chain_poses([(res[p].status, np.array(res[p].R), np.array(res[p].t)) for p in range(n - 1)], n)
    if dist is not None:
        all_traj = shard.gather_trajectories({rank: traj}, dist, cdev)
        torch.cuda.synchronize()
        assert len(all_traj) == world

    # ---- roofline: matcher kernel, HIP events on the library's stream (untimed passes) ----------
    # one chunk at a time on context 0, so that the matcher launch is timed without other streams
    # sharing the chip (the rocprofv3 summary in profiles/ is taken with --streams 1 for the same reason)
    nkp = [res[0].n_kp0] + [res[p].n_kp1 for p in range(n - 1)]
    nc = min(args.chunk, n - 1)  # pairs per matcher launch
    algo_bytes = sum(scan_bytes(nkp[p], nkp[p + 1]) for p in range(nc))
    ctx.prof_enable(True)
    for which in (0, 1, 2, 3):
        ctx.prof_get(which, True)
    for _ in range(5):
        ctx.pairs_run_dev(d_frames.data_ptr(), nc + 1, w, h, w, w * h, K, params)
    knn_n, knn_ms = ctx.prof_get(0)
    fast_n, fast_ms = ctx.prof_get(1)
    solve_n, solve_ms = ctx.prof_get(2)
    feat_n, feat_ms = ctx.prof_get(3)
    ctx.prof_enable(False)
    # detector + descriptor stage (pyramid .. rBRIEF) against its algorithmic bytes (SURVEY.md 8d:
    # W*H + 4*P + K*60 per frame, P = pyramid pixels): HBM-bound in the limit, VALU / latency bound today
    pyr_px = sum(int(round(w / 1.2 ** l)) * int(round(h / 1.2 ** l)) for l in range(8))
    feat_bytes = sum(w * h + 4 * pyr_px + k * 60 for k in nkp[: nc + 1])
    feat_s = feat_ms / max(feat_n, 1) * 1e-3
    detector = {"stage": "pyramid + FAST/NMS + selection + IC angle + blur + rBRIEF, %d frames per batch" % (nc + 1),
                "bound": "hbm", "achieved": feat_bytes / feat_s / 1e9 if feat_s > 0 else 0.0, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": feat_bytes / feat_s / 1e9 / HBM_PEAK_GBS if feat_s > 0 else 0.0,
                "algorithmic_bytes_per_batch": feat_bytes, "avg_batch_us": feat_s * 1e6}
    knn_avg_s = knn_ms / max(knn_n, 1) * 1e-3
    achieved = algo_bytes / knn_avg_s / 1e9 if knn_avg_s > 0 else 0.0
    traffic = None
    # HBM bytes per launch from the committed PMC passes (tools/gpu_profile.sh -> profiles/)
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_path):
        try:
            kern = json.load(open(pmc_path))["kernels"]
            traffic = [v for k, v in kern.items() if k.startswith("k_knn2_partial")][0]["hbm_bytes"]
        except Exception:
            traffic = None
    roofline = {"kernel": "k_knn2_partial<8, 1>", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "model": "O(N^2) scan model: nq*nt*32 + nq*32 + nq*16 bytes per pair (SURVEY.md 8d)",
                "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_us": knn_avg_s * 1e6,
                "pairs_per_launch": nc}
    # The tiles live in LDS, so the scan-model figure exceeds the HBM peak; the kernel's real ceiling is the
    # VALU one: 8 xor + 8 half-rate bcnt + key pack + min/max/min = 28 issue slots per
    # descriptor pair at the measured ~76 T lane-ops/s of the chip -> ~2.7 T descriptor pairs/s.
    desc_pairs = sum(nkp[p] * nkp[p + 1] for p in range(nc))
    roofline["valu"] = {"achieved_Tpairs_s": desc_pairs / knn_avg_s / 1e12, "ceiling_Tpairs_s": 2.7,
                        "frac": desc_pairs / knn_avg_s / 2.7e12}

    # ---- the same steps with the frames in pinned host memory: every chunk crosses PCIe inside the timed
    # region (the copy of one chunk overlaps the kernels of the others).  Reported beside `value`, never as it.
    pcie = None
    if rank == 0 and world == 1:
        pinned = torch.from_numpy(frames).pin_memory()
        seq.run(pinned.data_ptr(), n, w, h, w, w * h, K, params, mem=api.MEM_HOST_PINNED)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for _ in range(args.steps):
            seq.run(pinned.data_ptr(), n, w, h, w, w * h, K, params, mem=api.MEM_HOST_PINNED)
        tp = time.perf_counter() - tp
        pcie = {"value": args.steps * pairs_per_step / tp, "unit": "frame-pairs/s",
                "note": "frames in pinned host memory, uploads inside the timed region"}
        del pinned

    # ---- 8k x 8k matcher (BASELINE config 3 shape), same accounting -----------------------------
    matcher_8k = None
    if rank == 0 and not args.no_8k:
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
        b8 = scan_bytes(8192, 8192)
        matcher_8k = {"avg_launch_us": avg * 1e6, "achieved_GBs": b8 / avg / 1e9,
                      "frac_of_hbm_peak": b8 / avg / 1e9 / HBM_PEAK_GBS,
                      "algorithmic_bytes": b8, "compulsory_bytes": (8192 + 8192) * 32 + 8192 * 16}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(frames, K, args.nfeatures, args.max_iters)

    if rank == 0:
        ok = sum(1 for p in range(n - 1) if res[p].status == 0)
        line = {
            "metric": "frame-pairs/sec (1241x376, 2k ORB)",
            "value": total_pairs / elapsed,
            "unit": "frame-pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8 features/descriptors, u64 popcount matching, f64 geometry",
            "data": "synthetic (seeded planar-world KITTI-shaped frames rendered by visual_odometry_amd/synth.py)",
            "config": {"workload": "configs[1] batched: %d consecutive 1241x376 frames per GPU resident in HBM, "
                                   "ORB nfeatures=%d + BF-Hamming knn2/ratio/gate + 5-pt RANSAC + recoverPose per "
                                   "consecutive pair; chunks of %d pairs on %d HIP streams"
                                   % (n, args.nfeatures, args.chunk, args.streams),
                       "frames_per_step": n, "pairs_per_step": pairs_per_step, "nfeatures": args.nfeatures,
                       "ransac_max_iters": args.max_iters,
                       "streams": args.streams, "pairs_per_chunk": args.chunk,
                       "parallelism": "sequence shards x%d, no data-path collective" % world},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "matcher_8k": matcher_8k,
            "detector_roofline": detector,
            "pcie_inclusive": pcie,
            "stage_us_per_step": {"fast_score_kernel": fast_ms / max(fast_n, 1) * 1e3 if fast_n else None,
                                  "knn2_partial_kernel": knn_avg_s * 1e6,
                                  "solve5_kernels_per_step": solve_ms / 5 * 1e3},
            "host_ms_last_step": {"chunks": seq_stats[0], "queueing": seq_stats[1], "waiting": seq_stats[2]},
            "pairs_with_pose": ok,
            "keypoints_per_frame": float(np.mean(nkp)),
        }
        print(json.dumps(line))
    seq.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
