"""The multi-GPU driver on two CPU ranks (gloo): partitioning, the table broadcast, the
H-grid all-gather with uneven shards, pair round-robin and the final gather.  The
compute is the oracle injected as ``solve_fn`` - the HIP engine cannot run here, and the
point of these tests is the exchange logic."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_solve(table, denorm, vertices, gamma, sigma):
    """CPU stand-in with the ``solve_fn`` contract, computed from the broadcast table
    alone (so a rank that did not receive the table gets the wrong answer)."""
    t = table.cpu().numpy()
    den = denorm.cpu().numpy().reshape(4, 3, 3)
    v = vertices.cpu().numpy()
    if len(v) == 0:
        return torch.zeros((0, 9), dtype=torch.float32)
    from test_host_logic import MOMENT_INDEX
    P = np.zeros((len(t), 9, 9))
    for k, (i, j) in enumerate(MOMENT_INDEX):
        P[:, i, j] = t[:, k]
        P[:, j, i] = t[:, k]
    P[:, 3:6, 3:6] = P[:, 0:3, 0:3]
    s = t[:, 30:32]
    d = np.sqrt((v[:, None, 0] - s[None, :, 0]) ** 2 + (v[:, None, 1] - s[None, :, 1]) ** 2)
    w = np.exp(-(d * (1.0 / sigma ** 2)))
    w[w < gamma] = gamma
    M = ((w * w) @ P.reshape(len(t), 81)).reshape(-1, 9, 9)
    _, vec = np.linalg.eigh(M)
    h = vec[:, :, 0].reshape(-1, 3, 3)
    h = den[2] @ (den[0] @ h @ den[1]) @ den[3]
    h = h / h[:, 2:3, 2:3]
    return torch.from_numpy(h.astype(np.float32).reshape(-1, 9))


def oracle_warp_rows(img, H, mesh_w, mesh_h, final_w, final_h, off_x, off_y, row_begin, row_count, out_band, shape):
    """CPU stand-in with the ``warp_fn`` contract (warps the whole canvas, keeps the band)."""
    from oracle import apap_oracle as O
    rows, cols = shape
    Hn = H.cpu().numpy().reshape(rows, cols, 3, 3)
    hinv = np.linalg.inv(Hn.astype(np.float64)).astype(np.float32)
    full = O.local_warp_fast(img.cpu().numpy(), hinv, (mesh_w.cpu().numpy(), mesh_h.cpu().numpy()),
                             (final_w, final_h), (off_x, off_y))
    out_band[:row_count].copy_(torch.from_numpy(full[row_begin:row_begin + row_count]))
    return None


def oracle_warp_batch(imgs, H, mesh_w, mesh_h, final_w, final_h, off_x, off_y, shape, **kw):
    """CPU stand-in with the contract of ``hip_warp_batch`` (one oracle warp per pair of the batch)."""
    from oracle import apap_oracle as O
    rows, cols = shape
    out = torch.zeros((H.shape[0], final_h, final_w, 3), dtype=torch.uint8)
    for k in range(H.shape[0]):
        hinv = np.linalg.inv(H[k].numpy().reshape(rows, cols, 3, 3).astype(np.float64)).astype(np.float32)
        img = imgs[k] if imgs.dim() == 4 else imgs
        out[k] = torch.from_numpy(O.local_warp_fast(img.numpy(), hinv, (mesh_w.numpy(), mesh_h.numpy()), (final_w, final_h),
                                                    (off_x, off_y)))
    return out, None


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def worker(rank, world, port, rows, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cvx_proj_amd.dist import ShardedSolver, solve_pairs
        from cvx_proj_amd.synth import synth_pair
        p = synth_pair(640, 480, 150, rows, seed=21)        # rows x rows mesh
        p.vertices = p.vertices[:, :rows + 2] if rows > 3 else p.vertices   # non-square is fine
        if rank != 0:
            p.img = None                                        # only rank 0 owns the source image
        s = ShardedSolver(p, torch.device("cpu"), dist, solve_fn=oracle_solve, warp_fn=oracle_warp_rows)
        if rank != 0:
            assert float(s.table.abs().sum()) == 0.0            # only rank 0 holds the table before solve()
        assert s.overlap                                         # two launches per rank, the first gather beside the second
        H = s.solve().numpy().copy()
        assert float(s.table.abs().sum()) > 0.0                  # broadcast arrived
        s1 = ShardedSolver(p, torch.device("cpu"), dist, solve_fn=oracle_solve, warp_fn=oracle_warp_rows, overlap=False)
        assert not s1.overlap and np.array_equal(s1.solve().numpy(), H)      # one launch + one gather: the same grid
        assert np.array_equal(s.solve().numpy(), H)               # and again: the buffers are reused
        canvas = s.warp().numpy().copy()                         # image broadcast + banded warp + all-gather
        assert s.bands[0][0] == 0 and s.bands[-1][1] == p.final_h
        lo, hi = s.bands[rank]
        assert np.array_equal(s.warp(gather=False).numpy(), canvas[lo:hi])   # the canvas left distributed: own rows only
        # bands follow the mesh rows each rank solved: the warp reads the rank's OWN rows of the grid
        assert s._aligned and [b for b in s.bands if b[1] > b[0]][0][0] == 0
        H_step, band = s.step()                                  # solve, gather in flight, warp own band, wait
        assert np.array_equal(H_step.numpy(), H) and np.array_equal(band.numpy(), canvas[lo:hi])
        s.solve(wait=False)
        assert len(s._pending) > 0
        assert np.array_equal(s.finish().numpy(), H) and s._pending == []
        pairs = [synth_pair(320, 240, 60, 4, seed=100 + k) for k in range(5)]
        grids = solve_pairs(pairs, torch.device("cpu"), dist, solve_fn=oracle_solve)
        q.put((rank, s.parts, H, grids, canvas))
    except Exception as e:          # surface the failure instead of letting the parent time out
        import traceback
        q.put((rank, "ERROR", traceback.format_exc(), None, None))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("rows", [5, 8, 1])
def test_sharded_solver_two_ranks(rows):
    from oracle import apap_oracle as O
    from cvx_proj_amd.synth import synth_pair
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, rows, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    for r in res:
        assert r[1] != "ERROR", r[2]
    assert all(p.exitcode == 0 for p in procs)
    p = synth_pair(640, 480, 150, rows, seed=21)
    verts = p.vertices[:, :rows + 2] if rows > 3 else p.vertices
    H_ref, _ = O.local_homography_fast(p.src, p.dst, verts, p.gamma, p.sigma)
    hinv_ref = np.linalg.inv(H_ref.astype(np.float64)).astype(np.float32)
    canvas_ref = O.local_warp_fast(p.img, hinv_ref, (p.mesh[0], p.mesh[1]), (p.final_w, p.final_h), (p.off_x, p.off_y))
    for rank, parts, H, grids, canvas in res:
        # the warp ran on each rank's own H; the H grids equal the reference's to rounding, so
        # the canvases must agree except where a coordinate sits on an integer boundary
        assert np.mean((canvas != canvas_ref).any(axis=-1)) < 1e-4
        assert canvas.any()
        assert parts == res[0][1] and parts[0][0] == 0 and parts[-1][1] == verts.shape[0]
        H = H.reshape(verts.shape[0], verts.shape[1], 3, 3)
        assert O.reprojection_rmse_delta(H, H_ref, p.src).max() < 1e-6     # full grid on EVERY rank
    assert np.array_equal(res[0][2], res[1][2])
    assert np.array_equal(res[0][4], res[1][4])                  # same canvas on every rank
    assert res[1][3] is None
    grids = res[0][3]
    assert len(grids) == 5
    for k, g in enumerate(grids):
        pk = synth_pair(320, 240, 60, 4, seed=100 + k)
        ref, _ = O.local_homography_fast(pk.src, pk.dst, pk.vertices, pk.gamma, pk.sigma)
        assert O.reprojection_rmse_delta(g, ref, pk.src).max() < 1e-6, k


def worker8(rank, world, port, q):
    """Eight ranks: a 400-row mesh subsampled to 8 k rows and to an uneven 8 k + 3, both forms of the gather, the pipelined
    step; then BASELINE config 5's shape - 64 independent pairs dealt over 8 ranks, solved and warped."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    try:
        from cvx_proj_amd.dist import ShardedSolver, solve_pairs, warp_pairs
        from cvx_proj_amd.synth import synth_pair
        res = {}
        for rows in (16, 19):
            p = synth_pair(320, 240, 80, rows, seed=33)
            p.vertices = p.vertices[:, :5]                       # rows x 5 cells: cheap for the oracle
            p.mesh = np.stack([p.mesh[0], p.mesh[1]])
            if rank != 0:
                p.img = None
            grids = []
            for overlap in (False, True):
                s = ShardedSolver(p, torch.device("cpu"), dist, solve_fn=oracle_solve, warp_fn=oracle_warp_rows, overlap=overlap)
                assert s.overlap == overlap and s.world == 8
                grids.append(s.solve().numpy().copy())
            assert np.array_equal(grids[0], grids[1])
            assert not ShardedSolver(p, torch.device("cpu"), dist, solve_fn=oracle_solve).overlap      # "auto" at 8 ranks: one launch
            res[rows] = (s.parts, grids[0])
        pairs = [synth_pair(96, 64, 24, 3, seed=400 + k) for k in range(64)]
        grids = solve_pairs(pairs, torch.device("cpu"), dist, solve_fn=oracle_solve)
        # every rank needs the grids of ITS pairs for the warp: rank 0 scatters what it gathered
        box = [grids]
        dist.broadcast_object_list(box, src=0)
        mine = warp_pairs(pairs, box[0], torch.device("cpu"), dist, warp_fn=oracle_warp_batch)
        assert sorted(mine) == list(range(rank, 64, 8))
        allc = warp_pairs(pairs, box[0], torch.device("cpu"), dist, warp_fn=oracle_warp_batch, gather=True)
        if rank == 0:
            for k in mine:
                assert np.array_equal(allc[k], mine[k].numpy())
        q.put((rank, res, grids, allc))
    except Exception:
        import traceback
        q.put((rank, "ERROR", traceback.format_exc(), None))
        raise
    finally:
        dist.destroy_process_group()


def test_eight_ranks_cells_and_pairs():
    """What the driver's 8-GPU run executes, rehearsed on 8 CPU ranks: C4-style row shards (even and uneven, one launch or
    two per rank) and C5-style pairs (64 over 8 ranks, solved and warped), against the oracle."""
    from oracle import apap_oracle as O
    from cvx_proj_amd.synth import synth_pair
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(120)
    for r in res:
        assert r[1] != "ERROR", r[2]
    assert all(p.exitcode == 0 for p in procs)
    for rows in (16, 19):
        p = synth_pair(320, 240, 80, rows, seed=33)
        verts = p.vertices[:, :5]
        H_ref, _ = O.local_homography_fast(p.src, p.dst, verts, p.gamma, p.sigma)
        parts = res[0][1][rows][0]
        assert parts[0][0] == 0 and parts[-1][1] == rows and len(parts) == 8
        assert max(b - a for a, b in parts) - min(b - a for a, b in parts) == (0 if rows == 16 else 1)
        for r in res:
            H = r[1][rows][1].reshape(rows, 5, 3, 3)
            assert O.reprojection_rmse_delta(H, H_ref, p.src).max() < 1e-6          # the whole grid on every rank
            assert np.array_equal(r[1][rows][1], res[0][1][rows][1])
    grids, canvases = res[0][2], res[0][3]
    assert len(grids) == 64 and len(canvases) == 64 and all(r[2] is None and r[3] is None for r in res[1:])
    for k in (0, 7, 8, 37, 63):
        pk = synth_pair(96, 64, 24, 3, seed=400 + k)
        ref, _ = O.local_homography_fast(pk.src, pk.dst, pk.vertices, pk.gamma, pk.sigma)
        assert O.reprojection_rmse_delta(grids[k], ref, pk.src).max() < 1e-6, k
        hinv = np.linalg.inv(grids[k].astype(np.float64)).astype(np.float32)
        want = O.local_warp_fast(pk.img, hinv, pk.mesh, (pk.final_w, pk.final_h), (pk.off_x, pk.off_y))
        assert np.array_equal(canvases[k], want) and want.any(), k


def test_warp_pairs_refuses_mixed_geometry():
    from cvx_proj_amd.dist import warp_pairs
    from cvx_proj_amd.synth import synth_pair
    a, b = synth_pair(96, 64, 24, 3, seed=1), synth_pair(128, 64, 24, 3, seed=2)
    with pytest.raises(ValueError):
        warp_pairs([a, b], [None, None], torch.device("cpu"), warp_fn=oracle_warp_batch)


def test_row_partition():
    from cvx_proj_amd.dist import row_partition
    assert row_partition(400, 8) == [(50 * r, 50 * r + 50) for r in range(8)]
    assert row_partition(5, 2) == [(0, 3), (3, 5)]
    assert row_partition(1, 2) == [(0, 1), (1, 1)]
    parts = row_partition(203, 8)
    assert parts[0][0] == 0 and parts[-1][1] == 203 and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
    assert max(b - a for a, b in parts) - min(b - a for a, b in parts) <= 1


def test_hip_solve_refuses_cpu_tensors():
    from cvx_proj_amd import _native
    from cvx_proj_amd.dist import hip_solve
    with pytest.raises(_native.ApapError):
        hip_solve(torch.zeros((4, 32), dtype=torch.float64), torch.zeros(36, dtype=torch.float64),
                  torch.zeros((2, 2), dtype=torch.float64), 0.5, 100.0)


def test_same_bits_leaves_the_caller_s_context_alone_and_short_meshes_keep_unaligned_bands():
    """ShardedSolver(same_bits=True) pins APAP_OPT_PLAN_CELLS on a context of its OWN (the caller's keeps planning from its
    own calls); bands are aligned to the mesh rows only when the row edges reach the canvas's last row (edges that stop short must
    keep the index error of the single-GPU path, apap.py:207) - ADVICE r3."""
    from cvx_proj_amd import _native
    from cvx_proj_amd.dist import ShardedSolver, hip_solve
    from cvx_proj_amd.synth import synth_pair
    p = synth_pair(320, 240, 60, 6, seed=9, with_image=False)
    ctx = _native.Context(careful=0)
    try:
        s = ShardedSolver(p, torch.device("cpu"), None, solve_fn=hip_solve, ctx=ctx, same_bits=True)
        assert ctx.get("plan_cells") == 0 and ctx.get("careful") == 0
        assert s._ctx is not ctx and s._ctx.get("plan_cells") == 36 and s._ctx.get("careful") == 0
    finally:
        ctx.close()

    class FakeDist:
        def get_rank(self): return 1
        def get_world_size(self): return 2
        def broadcast(self, t, src=0): pass
    for short in (False, True):
        q = synth_pair(320, 240, 60, 6, seed=9)
        if short:
            q.mesh = q.mesh.copy()
            q.mesh[1, -1] = q.final_h - 5.0        # the last row edge stops 5 rows short of the canvas
        s = ShardedSolver(q, torch.device("cpu"), FakeDist(), solve_fn=oracle_solve, warp_fn=oracle_warp_rows)
        s._warp_setup()
        assert s._aligned == (not short)
        assert s.bands[0][0] == 0 and s.bands[-1][1] == q.final_h


def test_one_rank_rehearsal_runs_every_collective():
    """dist.REHEARSE_ONE_RANK: a group of ONE rank that does not take the single-process shortcuts - table broadcast, split H
    all-gather (both forms), image broadcast, aligned band + canvas all-gather, the pairs' gathers all run through the backend
    (here gloo; tests/test_gpu_dist.py runs the same switch over nccl = RCCL on the one-GPU box) - and gives the
    single-process results."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cvx_proj_amd import dist as D
    from cvx_proj_amd.synth import synth_pair
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    calls = []
    real = {name: getattr(dist, name) for name in ("broadcast", "all_gather_into_tensor", "gather")}
    try:
        for name, fn in real.items():
            setattr(dist, name, (lambda fn, name: lambda *a, **k: (calls.append(name), fn(*a, **k))[1])(fn, name))
        p = synth_pair(320, 240, 80, 6, seed=9)
        cpu = torch.device("cpu")
        plain = D.ShardedSolver(p, cpu, None, solve_fn=oracle_solve, warp_fn=oracle_warp_rows)
        H0, canvas0 = plain.solve().numpy().copy(), plain.warp().numpy().copy()
        assert calls == []
        D.REHEARSE_ONE_RANK = True
        for overlap in (True, False):
            calls.clear()
            s = D.ShardedSolver(p, cpu, dist, solve_fn=oracle_solve, warp_fn=oracle_warp_rows, overlap=overlap)
            assert s.overlap == overlap
            assert np.array_equal(s.solve().numpy(), H0)
            assert np.array_equal(s.warp().numpy(), canvas0) and s._aligned
            H_step, band = s.step()
            assert np.array_equal(H_step.numpy(), H0) and np.array_equal(band.numpy(), canvas0)
            assert calls.count("broadcast") == 3 and calls.count("all_gather_into_tensor") >= (2 if overlap else 1) + 1, calls
        pairs = [synth_pair(160, 120, 40, 3, seed=50 + k) for k in range(3)]
        calls.clear()
        grids = D.solve_pairs(pairs, cpu, dist, solve_fn=oracle_solve)
        canv = D.warp_pairs(pairs, grids, cpu, dist, warp_fn=oracle_warp_batch, gather=True)
        assert "gather" in calls
        D.REHEARSE_ONE_RANK = False
        grids0 = D.solve_pairs(pairs, cpu, None, solve_fn=oracle_solve)
        canv0 = D.warp_pairs(pairs, grids0, cpu, None, warp_fn=oracle_warp_batch, gather=True)
        assert all(np.array_equal(a, b) for a, b in zip(grids, grids0)) and all(np.array_equal(a, b) for a, b in zip(canv, canv0))
    finally:
        D.REHEARSE_ONE_RANK = False
        for name, fn in real.items():
            setattr(dist, name, fn)
        dist.destroy_process_group()
