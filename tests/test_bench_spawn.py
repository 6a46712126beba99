"""bench.py's N > 1 launch path on CPU: `python bench.py --gpus N` without a launcher starts N ranks
itself (the contract's torch.distributed.run line), refuses to run with fewer ranks or devices than
asked for, and — with --dry-run — moves clouds through the C ABI's exchange layout over gloo."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _run(*args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], cwd=ROOT, env=e,
                          capture_output=True, text=True, timeout=300)


def test_gpus2_dry_run_spawns_two_ranks_and_checks_the_exchange():
    r = _run("--gpus", "2", "--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["dry_run"] and d["ok"] and d["n_gpus"] == 2 and d["points"] > 0
    # ... and the gather to root (bench.py --exchange gather: rplgpu_gather_clouds_dev's layout)
    r = _run("--gpus", "2", "--dry-run", "--exchange", "gather")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["dry_run"] and d["ok"] and d["exchange"] == "gather"


def test_rank_count_mismatch_is_an_error():
    r = _run("--gpus", "2", "--dry-run", env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_more_gpus_than_devices_is_an_error():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = _run("--gpus", str(have + 1) if have else "2")
    assert r.returncode != 0 and "HIP device(s) are visible" in r.stderr
