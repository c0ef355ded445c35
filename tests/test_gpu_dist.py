"""The multi-rank path with the real HIP engine: two and three ranks run
ShardedSolver.solve() / warp() and solve_pairs; with same_bits the gathered grid equals the single-GPU one
bit for bit, by default to float64 summation order; the canvas equals the single-GPU warp of that grid.  On a one-GPU box the ranks share the device and talk over gloo (RCCL refuses two
ranks on one device); on a multi-GPU node the same program uses nccl = RCCL.  bench.py's own
`--gpus 2` launch is rehearsed the same way."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from cvx_proj_amd.synth import config_pair

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def clean_env(**extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


@pytest.mark.parametrize("world", [1, 2, 3])
def test_sharded_solver_ranks_equal_single_gpu(native, tmp_path, world):
    """world = 1 runs over nccl = RCCL (one rank per device: every collective of ShardedSolver, step()'s asynchronous gather and
    solve_pairs' gather go through RCCL once); 2 and 3 ranks share the one GPU of the box and talk over gloo."""
    out = str(tmp_path / "ranks.npz")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "tests", "_dist_gpu_ranks.py"), out, "C2"]
    r = subprocess.run(cmd, env=clean_env(), capture_output=True, text=True, timeout=600)      # a child, never exec
    assert r.returncode == 0, r.stderr[-3000:]
    z = np.load(out)
    p = config_pair("C2")
    from oracle import apap_oracle as O
    H, _ = native.local_homography(p.src, p.dst, p.vertices, p.gamma, p.sigma, want_weights=False)
    # same_bits: every shard sums its cells' keypoints in the whole mesh's order - the single-GPU grid bit for bit
    assert np.array_equal(z["H_same_bits"].reshape(H.shape), H)
    # default: a shard is a smaller launch and may take the fused kernel or other keypoint splits - the float64 sums
    # are grouped differently, a float32 value in thousands may round the other way (none does on C2 today)
    Hd = z["H"].reshape(H.shape)
    assert np.mean(Hd != H) < 1e-3 and O.reprojection_rmse_delta(Hd, H, p.src[:128]).max() < 1e-6
    canvas, _ = native.local_warp(p.img, Hd, p.mesh[0], p.mesh[1], p.final_w, p.final_h, p.off_x, p.off_y)
    assert np.array_equal(z["canvas"], canvas)                            # row bands + all-gather = the single-GPU canvas
    assert int(z["status"]) == 0
    parts, bands = z["parts"], z["bands"]
    assert len(parts) == world and parts[0][0] == 0 and parts[-1][1] == 100
    assert len(bands) == world and bands[0][0] == 0 and bands[-1][1] == p.final_h
    for k in range(5):                                                    # pairs dealt round-robin, gathered on rank 0
        pk = config_pair("C1", with_image=False, seed_offset=k)
        Hk, _ = native.local_homography(pk.src, pk.dst, pk.vertices, pk.gamma, pk.sigma, want_weights=False)
        assert np.array_equal(z["grids"][k], Hk)


def test_bench_self_launch_reports_pairs_and_cells(native):
    """`python bench.py --gpus 2` (no torchrun around it): both sharding modes in the one JSON line,
    each with the world size it saw; the cells leg's roofline is priced on rank 0's own shard."""
    env = clean_env(APAP_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--config", "C2", "--cells-config", "C2", "--no-cpu-baseline", "--c5-pairs", "6", "--cold-mb", "0"],
                       env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["mode"] == "pairs"
    assert d["pair_per_rank"]["world_size"] == 2 and d["pair_per_rank"]["value"] == d["value"]
    c5 = d["pairs"]                                                   # BASELINE config 5: pairs dealt over the ranks, batched
    assert c5["world_size"] == 2 and c5["scaling"] == "strong" and c5["pairs_per_rank"] == 3 and c5["value"] > 0
    assert c5["warp"]["value"] > 0 and c5["warp"]["us_per_pair"] > 0          # the warp half of config 5, batched per rank
    assert c5["whole_job"]["pairs_per_s"] > 0 and c5["whole_job"]["one_stream_ms_per_step"] > 0
    c = d["cells"]
    assert c["world_size"] == 2 and c["scaling"] == "strong" and c["backend"] == "gloo"
    assert c["rank0_cells"] == 50 * 100                               # half of the 100 x 100 mesh
    assert 0.0 < c["roofline"]["frac"] < 1.0 and c["value"] > 0 and c["warp"]["value"] > 0
    assert c["warp_bands_only"]["value"] >= c["warp"]["value"] * 0.9      # the all-gather can only cost
    assert c["collectives_overlapped"] is True and c["solve_other_form_ms_per_step"] > 0 and c["pipelined_step"]["ms_per_step"] > 0
    one = d["roofline"]["frac"]
    assert c["roofline"]["frac"] < 2.5 * one                          # not inflated by the world size


def test_bench_with_a_one_rank_rccl_group():
    """First contact with RCCL on a one-GPU box: bench.py with APAP_BENCH_FORCE_GROUP=1 builds an `nccl` process group of ONE rank and
    walks every multi-rank code path through it - the first all-reduce, the barriers, the table broadcast, the H all-gather and the
    canvas all-gather of `cells`, the max-over-ranks of the timings - with the real tensors (dtypes, contiguity, device).  One JSON
    line, the phase markers on stderr, the same canvases (bench.py asserts them)."""
    env = clean_env(APAP_BENCH_FORCE_GROUP="1")
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    env.pop("APAP_BENCH_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--condition-ms", "0", "--no-cpu-baseline",
                        "--no-call-level", "--cells-config", "C2", "--c5-pairs", "4"], env=env, capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["cells"]["value"] > 0 and d["pairs"]["warp"]["value"] > 0
    phases = [json.loads(ln)["bench_phase"] for ln in r.stderr.splitlines() if ln.startswith('{"bench_phase"')]
    assert phases[:2] == ["process group up", "first collective done"] and any(p.startswith("cells") for p in phases)
    assert all(json.loads(ln)["backend"] == "nccl" for ln in r.stderr.splitlines() if ln.startswith('{"bench_phase"'))


def test_single_process_sharded_solver_keeps_a_warp_plan(native):
    """No process group at all (what ``bench.py`` runs as `cells` on one GPU): the solver holds a WarpPlan over the whole mesh,
    the solve's tail leaves the cells warp ready, ``warp()`` / ``step()`` launch the gather kernel alone - and the canvas is the
    single-call one byte for byte."""
    import torch
    from cvx_proj_amd.dist import ShardedSolver
    dev = torch.device("cuda", 0)
    p = config_pair("C2")
    s = ShardedSolver(p, dev)
    H = s.solve().cpu().numpy().reshape(100, 100, 3, 3)
    assert s._plan is None                  # a caller that only solves gets the plain K2 and no canvas-sized workspace
    s = ShardedSolver(p, dev, resident_warp=True)
    assert np.array_equal(s.solve().cpu().numpy().reshape(100, 100, 3, 3), H)
    assert s._plan is not None and s._cells_ready and s._aligned and s.bands == [(0, p.final_h)]
    ctx = native.Context(profile=1)
    try:
        s2 = ShardedSolver(p, dev, ctx=ctx)
        s2.solve()
        s2.warp()
        ctx.profile_read()
        for _ in range(3):
            s2.warp()
        prof = ctx.profile_read()
        assert prof["warp"][1] == 3 and prof.get("invert", (0, 0))[1] == 0, prof       # three gathers, no set-up launch
        H2, canvas2 = s2.step()
        assert np.array_equal(H2.cpu().numpy().reshape(H.shape), H)
    finally:
        ctx.close()
    want, _ = native.local_warp(p.img, H, p.mesh[0], p.mesh[1], p.final_w, p.final_h, p.off_x, p.off_y)
    assert np.array_equal(s.warp().cpu().numpy(), want) and np.array_equal(canvas2.cpu().numpy(), want)
    assert int(s.status.cpu()[0]) == 0
    # a mesh whose row edges stop short of the canvas keeps the all-phase call and reports the reference's IndexError rows
    import copy
    q = copy.copy(p)
    q.mesh = p.mesh.copy()
    q.mesh[1] = p.mesh[1] * 0.5
    s3 = ShardedSolver(q, dev)
    s3.solve()
    s3.warp()
    assert s3._plan is None and not s3._aligned and int(s3.status.cpu()[0]) & 2
