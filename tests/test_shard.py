"""N > 1 path on the CPU: sequence sharding (LPT) and the trajectory gather that bench.py runs over
RCCL, here over gloo with world_size 2 (no data-path collective, gather only)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from visual_odometry_amd import shard


def test_lpt_assignment_balances_kitti_lengths():
    L = shard.KITTI_LENGTHS
    assert len(L) == 22 and L[0] == 4541 and L[4] == 271
    for world in (1, 2, 4, 8):
        a = shard.lpt_assign(L, world)
        assert sorted(i for r in a for i in r) == list(range(22))
        loads = [sum(L[i] for i in r) for r in a]
        # LPT is within 4/3 of the optimum; on these lengths it is far better than round-robin
        assert max(loads) <= max(sum(L) / world * 1.15, max(L))
    assert shard.lpt_assign(L, 8) == shard.lpt_assign(L, 8)


def test_chain_poses_convention():
    c, s = np.cos(0.1), np.sin(0.1)
    R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    t = np.array([0.0, 0.0, -1.0])
    traj = shard.chain_poses([(0, R, t), (4, R, t), (0, R, t)], 4)
    assert np.allclose(traj[0], np.hstack([np.eye(3), np.zeros((3, 1))]).reshape(-1))
    assert np.allclose(traj[1].reshape(3, 4)[:, 3], [0, 0, 1])  # t = -t at identity pose
    assert np.allclose(traj[2], traj[1])  # failed pair repeats the pose
    assert np.allclose(traj[3].reshape(3, 4)[:, :3], R.T @ R.T)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = shard.lpt_assign([5, 3, 4, 2, 1], world)[rank]
        local = {i: np.full((i + 2, 12), float(i)) + np.arange(12) for i in mine}
        allt = shard.gather_trajectories(local, dist, torch.device("cpu"))
        ok = sorted(allt) == [0, 1, 2, 3, 4] and all(
            allt[i].shape == (i + 2, 12) and np.array_equal(allt[i], np.full((i + 2, 12), float(i)) + np.arange(12))
            for i in allt)
        # the timing reduction bench.py uses: max over ranks
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        q.put((rank, ok and float(t.item()) == float(world)))
    finally:
        dist.destroy_process_group()


def test_gather_trajectories_gloo_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]
