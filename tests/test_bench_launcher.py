"""`python bench.py --gpus N` from a plain shell: the script starts its own ranks (one process per GPU), forwards rank 0's line and fails
loudly when a rank fails -- the way the reference launches its parallel tests (cmake/TestRunner.cmake:5-18: `${MPIEXEC} -n N exe`)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout):
    env = dict(os.environ, **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)


def test_launcher_reports_failed_ranks():
    """No GPU here (or a workload no rank can build): every rank fails, the launcher exits non-zero, says which rank, prints no result line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the ranks would succeed")
    r = _run(["--gpus", "2", "--workload", "lap7_8", "--steps", "2", "--warmup", "1", "--legs", "none"], {"PHIP_BENCH_TIMEOUT": "240"}, 300)
    assert r.returncode != 0
    assert "exited with" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,workload,ipc", [(2, "box27_48", "0"), (3, "box27_48_var", "1")])
def test_bench_starts_its_own_ranks(ranks, workload, ipc):
    """Rehearsal on one GPU (PHIP_BENCH_BACKEND=shm lets the ranks share it): the line carries what a SCALE run is read by."""
    r = _run(["--gpus", str(ranks), "--workload", workload, "--steps", "20", "--warmup", "3"], {"PHIP_BENCH_BACKEND": "shm", "PHIP_HALO_IPC": ipc, "PHIP_BENCH_TIMEOUT": "400"}, 500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["steps"] == 20 and d["scaling"] == "strong" and d["unit"] == "GB/s"
    mg = d["multi_gpu"]
    assert mg["comm_ranks"] == ranks and mg["backend"] == "shm" and mg["rccl_ranks"] == 0
    assert mg["matmult_check"]["ok"] and max(mg["matmult_check"]["max_rel_err_vs_numpy"]) <= 1e-12
    assert ("peer writes" in mg["ghost_exchange"]) == (ipc == "1")
    assert d["value"] <= 8000.0 * ranks and d["cg_iters_per_s"] > 0
    assert "legs" not in d and "cpu_baseline" not in d
