"""bench.py's own launcher and its CPU-only legs (no GPU needed): `--gpus N` without WORLD_SIZE must
start N ranks as child processes (never exec), rendezvous on 127.0.0.1 and let rank 0 print one JSON
line; the CPU-baseline rows must come out of a bounded sample."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("ranks", [2, 8])
def test_self_launch_ranks_gloo(ranks):
    """2 ranks, and the 8 the driver's scaling run starts (the launch, the rendezvous and one collective; no GPU)."""
    env = dict(os.environ, APAP_BENCH_SELFTEST="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                 # rank 0 only
    d = json.loads(lines[0])
    assert d["selftest"] and d["n_gpus"] == ranks and d["sum"] == ranks * (ranks + 1) / 2     # every rank contributed
    assert d["ranks_env"][2] == str(ranks) and d["ranks_env"][3] == "127.0.0.1"


def test_a_failing_rank_exits_non_zero_with_evidence():
    """A rank that cannot get a GPU (none here) must end the run with a non-zero code and say where it was - not hang,
    not print a JSON line."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present: the run succeeds")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "APAP_BENCH_SELFTEST"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--no-cpu-baseline"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "bench_failed" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]


def test_cpu_baseline_rows_on_a_small_config():
    sys.path.insert(0, ROOT)
    import bench
    out = bench.cpu_baseline("C1", budget_cells=40, budget_rows=4, pool_workers=2, default_threads_budget_s=1.0)
    assert out["cores"] == 1 and out["value"] > 0 and out["warp_value"] > 0
    assert out["all_cores"]["cores"] == 2 and out["all_cores"]["value"] > 0
    assert out["default_blas_threads"]["value"] > 0 and "error" not in out["default_blas_threads"]
