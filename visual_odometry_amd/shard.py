"""Multi-GPU sharding of the frame-pair path.

The path has no exchange step: frame pairs of different sequences are independent, so sequences
(or contiguous frame-pair batches of one sequence) are the shard unit, one process per GPU, and the
only communication is the gather of the finished trajectories (N x 12 doubles per sequence) and a
few counters to rank 0 — one padded all_gather over RCCL (backend "nccl" on ROCm) after the timed
region, tens of kilobytes, latency-bound.  Nothing in here touches the data path.
"""
import numpy as np

# KITTI odometry sequence lengths: 00-10 from the reference (src/kitti/readme.txt:125-135),
# 11-21 as recorded in SURVEY.md §8d.
KITTI_LENGTHS = [4541, 1101, 4661, 801, 271, 2761, 1101, 1101, 4071, 1591, 1201,
                 921, 1061, 3281, 631, 1901, 1731, 491, 1801, 4981, 831, 2721]


def lpt_assign(lengths, world):
    """Longest-processing-time-first assignment of sequences to `world` ranks by frame count.
    Returns a list (per rank) of sequence ids; deterministic (ties by lower id)."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    load = [0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += lengths[i]
    return out


def chain_poses(rel, n_frames):
    """Relative poses (status, R, t) of consecutive pairs -> n_frames x 12 camera-to-world trajectory,
    with the reference's convention (vo_pose.h:29-38: t = -t, R = R^T, pose_t += pose_R t,
    pose_R = R pose_R) at unit scale.  A failed pair repeats the last pose (vo_pose.h:45-48)."""
    traj = np.zeros((n_frames, 12))
    R_w, t_w = np.eye(3), np.zeros(3)
    traj[0] = np.hstack([R_w, t_w[:, None]]).reshape(-1)
    for p in range(n_frames - 1):
        status, R, t = rel[p]
        if status == 0:
            t_w = t_w + R_w @ (-np.asarray(t, dtype=np.float64))
            R_w = np.asarray(R, dtype=np.float64).reshape(3, 3).T @ R_w
        traj[p + 1] = np.hstack([R_w, t_w[:, None]]).reshape(-1)
    return traj


def gather_trajectories(local, dist, device):
    """local: {sequence id: (n, 12) float64 array} of this rank.  Every rank gets
    {sequence id: array} of all ranks (rank 0 is the consumer).  Two collectives: the per-rank
    table sizes, then one padded all_gather of [id, n, n x 12 doubles] records."""
    import torch

    world = dist.get_world_size()
    ids = sorted(local)
    rec = [np.concatenate([[float(i), float(len(local[i]))], np.asarray(local[i], np.float64).reshape(-1)]) for i in ids]
    flat = np.concatenate(rec) if rec else np.zeros(0)
    n_local = torch.tensor([flat.size], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    cap = max(int(s.item()) for s in sizes)
    buf = torch.zeros(max(cap, 1), dtype=torch.float64, device=device)
    if flat.size:
        buf[: flat.size] = torch.from_numpy(flat).to(device)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    out = {}
    for r in range(world):
        a = bufs[r].cpu().numpy()[: int(sizes[r].item())]
        o = 0
        while o < a.size:
            sid, n = int(a[o]), int(a[o + 1])
            out[sid] = a[o + 2: o + 2 + n * 12].reshape(n, 12).copy()
            o += 2 + n * 12
    return out
