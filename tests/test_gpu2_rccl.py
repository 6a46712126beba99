"""Real RCCL ranks (BASELINE configs 4 and 5 across GPUs).  A 1-GPU box can only
run the one-rank form (RCCL refuses two ranks on one device): the worker and the bench launch are
exercised there with world = 1, so that the two-rank tests — marked gpu2, skipped below two
devices — differ from something that ran by the number of ranks only."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _env(port):
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


def _run_worker(world, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "tests" / "rccl_c5_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=_env(port), cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert f"RCCL_C5_OK {world}" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def _run_bench(world, port):
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--scans", "512", "--cpu-seconds", "0", "--no-laserscan", "--no-variants", "--no-decode",
           "--no-single"]
    env = _env(port)
    if world == 1:
        env["RPL_BENCH_FORCE_DIST"] = "1"  # the N > 1 code path with a single rank
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-4000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == world and line["status_bits"] == 0 and line["value"] > 0
    assert line["exchange_backend"].startswith("rccl"), line["exchange_backend"]
    assert line["exchange_ranks"] == world  # ncclCommCount, not what was asked for
    assert line["compute_only_ms"] > 0 and line["exchange_only_ms"] > 0
    assert line["gathered_bytes_per_rank"] == line["compute_only"]["gathered_bytes_per_rank"]
    if world > 1:
        assert line["gathered_bytes_per_rank"] > 0
    return line


def test_c5_worker_one_rank():
    _run_worker(1, 29541)


def test_bench_exchange_path_one_rank():
    _run_bench(1, 29542)


@pytest.mark.gpu2
def test_c5_one_sensor_group_per_rank_fused_message_two_ranks():
    """Config 5 across two GPUs: four sensors per rank, the fused PointCloud2 on every rank equal
    to the single-GPU chain."""
    _run_worker(2, 29543)


@pytest.mark.gpu2
def test_bench_two_gpus():
    """Config 4 at two GPUs: bench.py --gpus 2 starts its own ranks, the communicator spans both."""
    _run_bench(2, 29544)
