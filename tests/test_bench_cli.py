"""bench.py's launch contract, without a GPU: `--gpus N` starts N ranks itself (before any GPU call), the
ranks shard the 22 KITTI-shaped sequences by LPT, meet at the barriers and gather the trajectories
(gloo here, RCCL on GPUs), and rank 0 prints one JSON line with n_gpus == N.  `--dry-run` replaces the HIP
work by identity poses: plumbing only."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True,
                          text=True, timeout=300)


def test_gpus_2_launches_two_ranks_and_reports_them():
    r = _run(["--gpus", "2", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["dry_run"] is True
    assert j["sequences"] == 22 and j["trajectories_gathered"] == 22
    assert j["pairs_per_step"] == 43552 - 22 and j["scaling"] == "strong"
    assert 1.0 <= j["lpt_imbalance"] < 1.01


def test_gpus_must_agree_with_the_launcher():
    r = _run(["--gpus", "4", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


def test_single_rank_dry_run():
    r = _run(["--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["pairs_per_step"] == 4540


import pytest  # noqa: E402


@pytest.mark.parametrize("world", [4, 8])
def test_gpus_4_and_8_shard_all_22_sequences(world):
    """The 1 -> 8 path without a node: N ranks over gloo, LPT shards within 2 % of perfect balance, every one
    of the 22 trajectories back on rank 0 with its own length (bench.py asserts the lengths before printing)."""
    r = _run(["--gpus", str(world), "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == world and j["sequences"] == 22 and j["trajectories_gathered"] == 22
    assert j["pairs_per_step"] == 43530 and j["scaling"] == "strong"
    assert 1.0 <= j["lpt_imbalance"] <= 1.02


def test_driver_launch_line_torch_distributed_run():
    """The driver's own command for N > 1: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N."""
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "1", "--warmup", "0", "--dry-run"], env=e, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_the_self_launcher_runs_before_anything_can_touch_the_gpu():
    """`bench.py --gpus N` starts its ranks as CHILD processes (never an exec of itself) and does so before
    torch, the HIP library or the package's bindings are imported: a parent that has initialised the GPU must
    not replace itself, and one that only spawns needs no GPU at all."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    launch = main.index("self_launch(args)")
    for needle in ("import torch", "from visual_odometry_amd import api", "api.Vo"):
        assert needle not in main[:launch], needle
    head = src[: src.index("def main():")]
    assert "import torch" not in head and "visual_odometry_amd" not in head.split('"""', 2)[2].split("def self_launch")[0]
    assert "os.exec" not in src and "execv" not in src
    # and it really does not: with no GPU visible the parent still launches its ranks (they run the dry path)
    r = _run(["--gpus", "2", "--dry-run"], {"HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""})
    assert r.returncode == 0, r.stderr[-2000:]
