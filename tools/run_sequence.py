#!/usr/bin/env python3
"""Runs one KITTI-format sequence through the accelerated front end and writes the trajectory.

Mirrors the reference's per-image loop (src/main.cpp:21-66,115: list and sort the images of
<data_dir>/sequences/<seq>/<image_dir>, read K from line <calib_line_number> of calib.txt, scale
it by image_scale, VisualOdometry::addImage per frame, collect the poses), with the flag names of
src/main.h:27-46.  Two pipelines:
  --pipeline lk    the reference's default: FAST -> pyramidal LK -> LMedS E -> recoverPose, through
                   the C++ host mirror's VisualOdometry class (one frame at a time, as the reference)
  --pipeline orb   the descriptor path of BASELINE.json: ORB -> BF-Hamming -> RANSAC E -> recoverPose
                   for every consecutive pair, batched through vo_seq_run (relative poses chained
                   with unit steps: no scale without the tracker's 3-D points)
Writes <res_dir>/<seq>.txt (12 doubles per line, the format of the ground-truth poses) and prints one
JSON line with the KITTI translation / rotation error when ground truth is present.
Drawing, plotting and the GTSAM back end (--ba) are not part of this path."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data_dir", required=True, help="KITTI odometry root (holds sequences/ and poses/)")
    ap.add_argument("--res_dir", required=True)
    ap.add_argument("--poses", default=None, help="ground-truth pose directory (default <data_dir>/poses)")
    ap.add_argument("--seq", default="00")
    ap.add_argument("--image_dir", default="image_0")
    ap.add_argument("--image_scale", type=float, default=1.0)
    ap.add_argument("--min_tracked_features", type=int, default=750)
    ap.add_argument("--calib_file", default="calib.txt")
    ap.add_argument("--calib_line_number", type=int, default=0)
    ap.add_argument("--pipeline", choices=["lk", "orb"], default="lk")
    ap.add_argument("--scale_seed", type=int, default=1234)
    ap.add_argument("--tracker_channels", type=int, default=3)
    ap.add_argument("--max_frames", type=int, default=0)
    args = ap.parse_args()

    from PIL import Image

    from visual_odometry_amd import host_api

    seq_dir = os.path.join(args.data_dir, "sequences", args.seq)
    img_dir = os.path.join(seq_dir, args.image_dir)
    names = sorted(f for f in os.listdir(img_dir) if os.path.isfile(os.path.join(img_dir, f)))
    if args.max_frames:
        names = names[: args.max_frames]
    K = host_api.load_calibration(os.path.join(seq_dir, args.calib_file), args.calib_line_number)
    K = K * args.image_scale
    K[2, 2] = 1.0

    def read(name):
        im = Image.open(os.path.join(img_dir, name)).convert("L")
        if args.image_scale != 1.0:
            im = im.resize((round(im.width * args.image_scale), round(im.height * args.image_scale)), Image.BILINEAR)
        return np.asarray(im, dtype=np.uint8)

    t0 = time.perf_counter()
    if args.pipeline == "lk":
        vo = host_api.HostVisualOdometry(K, args.min_tracked_features, args.scale_seed, args.tracker_channels)
        poses = np.stack([vo.add_image(read(n)) for n in names])
        vo.close()
    else:
        from visual_odometry_amd import api, shard

        frames = np.stack([read(n) for n in names])
        n, h, w = frames.shape
        seq = api.VoSequence(0, n_streams=3, pairs_per_chunk=32, max_width=max(w, 1241), max_height=max(h, 376))
        res = seq.run(frames, n, w, h, w, w * h, K, api.pair_params(2000), mem=api.MEM_HOST)
        poses = shard.chain_poses([(res[p].status, np.array(res[p].R), np.array(res[p].t)) for p in range(n - 1)], n)
        poses = poses.reshape(n, 3, 4)
        seq.close()
    dt = time.perf_counter() - t0
    os.makedirs(args.res_dir, exist_ok=True)
    out = os.path.join(args.res_dir, args.seq + ".txt")
    host_api.save_poses(out, poses)
    rep = {"seq": args.seq, "pipeline": args.pipeline, "frames": len(names), "seconds": dt,
           "frames_per_s": len(names) / dt, "result": out}
    gt_path = os.path.join(args.poses or os.path.join(args.data_dir, "poses"), args.seq + ".txt")
    gt = host_api.load_poses(gt_path)
    if len(gt):
        err = host_api.kitti_sequence_errors(gt[: len(poses)], poses)
        if err:
            rep.update(t_err_percent=100 * err[0], r_err_deg_per_100m=float(np.degrees(err[1]) * 100), segments=err[2])
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
