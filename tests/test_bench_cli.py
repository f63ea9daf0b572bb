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
