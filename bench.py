#!/usr/bin/env python3
"""Benchmark of the APAP hot path on MI355X: local homographies/s and warp Mpix/s on the
4K pair / 200x200 mesh / 2000 keypoints configuration of BASELINE.json (C3).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3] [--variant auto|valu|mfma]
                    [--no-cpu-baseline] [--mode pairs|cells]

A "step" is one pass of the hot path over one image pair with all inputs resident in
HBM: the per-cell solve (assemble + eigen-solve kernels) and the backward warp
(invert + lookup + gather kernels).  The two halves of the metric are timed in two
regions of exactly K steps each, both bracketed by a barrier and a device synchronise:
``value`` = cells * K * N / t_solve (homographies/s), ``warp.value`` = canvas pixels
* K * N / t_warp (Mpix/s), ``ms_per_step`` = (t_solve + t_warp) / K.

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): ``--mode pairs``
(default) gives every rank its own 4K pair - pairs are independent, no collective on
the data path, weak scaling.  ``--mode cells`` shards the mesh rows of ONE pair over
the ranks with a broadcast of the keypoint table and an all-gather of the H grid
(cvx_proj_amd.dist), strong scaling.

Extra objects on the JSON line: ``roofline`` for the dominant kernel (K1, timed live
with HIP events on the launch stream) and ``cpu_baseline`` (the oracle's
faithful-loop numpy port timed on this host, one thread, bounded sample).
"""
import os

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")   # the CPU baseline is the 1-thread port
os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import argparse
import ctypes
import json
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from cvx_proj_amd import _native as N  # noqa: E402
from cvx_proj_amd.synth import CONFIGS, config_pair  # noqa: E402

METRIC = "local homographies/sec + Mpix/sec warp, 4K pair 200x200 mesh"
# Peaks (DESIGN.md section "Rooflines"): fp64 matrix = fp64 vector = 78.6 TFLOP/s is AMD's
# MI355X datasheet figure (256 CU x 4 SIMD x 2.4 GHz x 32 flop/clk); HBM 8 TB/s spec from
# /opt/skills/guides/MI355X_MICROARCH.md.
PEAK_FP64_TFLOPS = 78.6
PEAK_HBM_GBS = 8000.0
# algorithmic flops per (cell, keypoint) of K1, SURVEY.md 8(d): 8 (weight) + 48 (24 FMAs) + 2
K1_FLOPS_PER_CELL_POINT = 58.0


class Resident:
    """One image pair (or a batch of pairs sharing mesh and image) with everything the hot
    path reads resident in HBM."""

    def __init__(self, pair, dev, batch=1, seed_pairs=None):
        self.pair = pair
        self.batch = batch
        q = N.host_prepare(pair.src, pair.dst)
        table = N.host_build_table(pair.src, q["cf1"], q["cf2"])
        den = N.host_build_denorm(q["iC2"], q["C1"], q["iN2"], q["N1"])
        if batch > 1:      # independent keypoint sets (different seeds), same mesh
            tabs, dens = [table], [den]
            for extra in seed_pairs:
                qe = N.host_prepare(extra.src, extra.dst)
                tabs.append(N.host_build_table(extra.src, qe["cf1"], qe["cf2"]))
                dens.append(N.host_build_denorm(qe["iC2"], qe["C1"], qe["iN2"], qe["N1"]))
            table, den = np.stack(tabs), np.stack(dens)
        self.n = len(pair.src)
        self.rows, self.cols = pair.vertices.shape[:2]
        self.cells = self.rows * self.cols
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        self.table, self.den = t(table), t(den)
        self.vert = t(pair.vertices.reshape(-1, 2))
        self.H = torch.zeros((batch * self.cells, 9), dtype=torch.float32, device=dev)
        self.work_bytes = max(N.lib().apap_solve_batch_workspace_bytes(self.n, self.cells, batch), 256)
        self.work = torch.empty(self.work_bytes, dtype=torch.uint8, device=dev)
        self.img = t(pair.img)
        self.mesh_w, self.mesh_h = t(pair.mesh[0]), t(pair.mesh[1])
        self.out = torch.zeros((pair.final_h, pair.final_w, 3), dtype=torch.uint8, device=dev)
        self.wwork_bytes = N.lib().apap_warp_workspace_bytes(self.rows, self.cols, pair.final_w, pair.final_h)
        self.wwork = torch.empty(self.wwork_bytes, dtype=torch.uint8, device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)

    def solve(self, stream):
        p = self.pair
        N.check(N.lib().apap_solve_batch_device(self.table.data_ptr(), self.n, self.vert.data_ptr(), 0, self.cells,
                                                p.gamma, p.sigma, self.den.data_ptr(), self.H.data_ptr(), self.batch,
                                                self.work.data_ptr(), self.work_bytes, ctypes.c_void_p(stream)))

    def warp(self, stream):
        p = self.pair
        N.check(N.lib().apap_warp_device(self.img.data_ptr(), p.shape[0], p.shape[1], self.H.data_ptr(), self.rows,
                                         self.cols, self.mesh_w.data_ptr(), p.mesh.shape[1], self.mesh_h.data_ptr(),
                                         p.mesh.shape[1], p.final_w, p.final_h, p.off_x, p.off_y,
                                         self.out.data_ptr(), None, self.wwork.data_ptr(), self.wwork_bytes,
                                         self.status.data_ptr(), ctypes.c_void_p(stream)))

    def equalize(self, stream):
        """Per-channel histogram equalisation of the source image (the pre-processing of
        apap.py:236-237); timed as an extra, not part of ``value``."""
        p = self.pair
        if not hasattr(self, "eq_out"):
            self.eq_out = torch.empty_like(self.img)
            self.eq_work = torch.zeros(N.lib().apap_equalize_workspace_bytes(3), dtype=torch.uint8, device=self.img.device)
        N.check(N.lib().apap_equalize_hist_device(self.img.data_ptr(), p.shape[0], p.shape[1], 3, self.eq_out.data_ptr(),
                                                  self.eq_work.data_ptr(), self.eq_work.numel(), ctypes.c_void_p(stream)))

    def ransac(self, stream):
        """Device half of the seed-homography estimator (baseline_stitch_test.py:42) on the pair's
        correspondences; timed as an extra, not part of ``value``."""
        p = self.pair
        if not hasattr(self, "r_src"):
            dev = self.img.device
            self.r_src = torch.from_numpy(np.ascontiguousarray(p.src, dtype=np.float32)).to(dev)
            self.r_dst = torch.from_numpy(np.ascontiguousarray(p.dst, dtype=np.float32)).to(dev)
            self.r_wb = N.lib().apap_ransac_workspace_bytes(self.n, N.RANSAC_ITERATIONS)
            self.r_work = torch.zeros(self.r_wb, dtype=torch.uint8, device=dev)
            self.r_H = torch.zeros(9, dtype=torch.float64, device=dev)
            self.r_mask = torch.zeros(self.n, dtype=torch.uint8, device=dev)
            self.r_res = torch.zeros(2, dtype=torch.int32, device=dev)
        N.check(N.lib().apap_ransac_device(self.r_src.data_ptr(), self.r_dst.data_ptr(), self.n, 5.0, N.RANSAC_ITERATIONS,
                                           ctypes.c_ulonglong(N.RANSAC_SEED), self.r_H.data_ptr(), self.r_mask.data_ptr(),
                                           self.r_res.data_ptr(), self.r_work.data_ptr(), self.r_wb,
                                           ctypes.c_void_p(stream)))

    def stitch(self, stream):
        """Fused warp + paste + uniform_blend (the reference's commented-out tail,
        apap.py:258-262); timed as an extra, not part of ``value``."""
        p = self.pair
        if not hasattr(self, "center"):
            g = torch.Generator(device="cpu").manual_seed(1)
            self.center = torch.randint(0, 256, p.shape, dtype=torch.uint8, generator=g).to(self.img.device)
        N.check(N.lib().apap_stitch_device(self.img.data_ptr(), p.shape[0], p.shape[1], self.center.data_ptr(),
                                           p.shape[0], p.shape[1], self.H.data_ptr(), self.rows, self.cols,
                                           self.mesh_w.data_ptr(), p.mesh.shape[1], self.mesh_h.data_ptr(),
                                           p.mesh.shape[1], p.final_w, p.final_h, p.off_x, p.off_y,
                                           self.out.data_ptr(), None, self.wwork.data_ptr(), self.wwork_bytes,
                                           self.status.data_ptr(), ctypes.c_void_p(stream)))


def cpu_baseline(cfg, budget_cells, budget_rows, pool_workers=0):
    """The oracle's faithful-loop port on this host: a bounded, seeded sample of the
    same workload (cells spread over the mesh; every k-th canvas row)."""
    from oracle import apap_oracle as O
    p = config_pair(cfg)
    rows, cols = p.vertices.shape[:2]
    rng = np.random.default_rng(0)
    flat = rng.choice(rows * cols, size=min(budget_cells, rows * cols), replace=False)
    cells = [(int(f // cols), int(f % cols)) for f in flat]
    t0 = time.perf_counter()
    H, _ = O.local_homography_loop(p.src, p.dst, p.vertices, p.gamma, p.sigma, cells=cells, want_weights=True)
    t_solve = time.perf_counter() - t0
    H_full, _ = O.local_homography_fast(p.src, p.dst, p.vertices, p.gamma, p.sigma)
    sub = list(range(0, p.final_h, max(1, p.final_h // budget_rows)))[:budget_rows]
    hinv = H_full.copy()
    t0 = time.perf_counter()
    O.local_warp_loop(p.img, hinv, p.mesh, (p.final_w, p.final_h), (p.off_x, p.off_y), rows_subset=sub)
    t_warp = time.perf_counter() - t0
    # the in-place inversion of all cells is part of local_warp; it is inside t_warp, as in the reference
    pool = None
    if pool_workers > 1:
        # "best-effort CPU" row of BASELINE.md: the same loop over a process pool, 1 BLAS thread each
        Hp, t_pool = O.local_homography_pool(p.src, p.dst, p.vertices, p.gamma, p.sigma, cells, pool_workers)
        assert np.array_equal(Hp, H)
        pool = {"value": len(cells) / t_pool, "unit": "homographies/s", "cores": pool_workers, "kind": "port",
                "sample": f"same {len(cells)} cells over a {pool_workers}-process pool (1 BLAS thread each; worker "
                          f"start-up excluded), {t_pool:.2f} s"}
    return {
        "pool": pool,
        "value": len(cells) / t_solve, "unit": "homographies/s", "cores": 1, "kind": "port",
        "sample": f"{len(cells)} of {rows * cols} cells (seeded random), {len(sub)} of {p.final_h} canvas rows + all "
                  f"{rows * cols} cell inversions; oracle faithful-loop numpy port, OPENBLAS_NUM_THREADS=1, "
                  f"host has {os.cpu_count()} logical cores",
        "warp_value": len(sub) * p.final_w / t_warp / 1e6, "warp_unit": "Mpix/s",
        "solve_s": t_solve, "warp_s": t_warp,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C3", choices=sorted(CONFIGS))
    ap.add_argument("--variant", default="auto", choices=["auto", "valu", "mfma", "mfma4", "mfma4x2"])
    ap.add_argument("--mode", default="pairs", choices=["pairs", "cells"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=1,
                    help="solve this many independent pairs per step in ONE batched launch (config C5 style); "
                         "the warp half then runs once per pair")
    ap.add_argument("--graph", action="store_true",
                    help="capture the solve and the warp step into HIP graphs and time graph replays")
    ap.add_argument("--cpu-cells", type=int, default=40000)
    ap.add_argument("--cpu-rows", type=int, default=400)
    ap.add_argument("--cpu-pool", type=int, default=16, help="workers of the process-pool CPU row (0 = skip)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world
    # one rank per GPU; if a rehearsal runs more ranks than GPUs (e.g. 2 gloo ranks on a 1-GPU
    # box, APAP_BENCH_BACKEND=gloo) the ranks share devices round-robin
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("APAP_BENCH_BACKEND", "nccl")      # nccl = RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    else:
        dist = None

    N.lib().apap_set_solver_variant({"auto": 0, "valu": 1, "mfma": 2, "mfma4": 3, "mfma4x2": 4}[a.variant])
    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()
        torch.cuda.synchronize()

    if a.mode == "cells" and world > 1:
        from cvx_proj_amd.dist import ShardedSolver
        pair = config_pair(a.config)
        res = ShardedSolver(pair, dev, dist)
        units_solve = res.cells_total                # strong scaling: one pair for the whole job
        units_warp = pair.final_w * pair.final_h
        scaling = "strong"
    else:
        pair = config_pair(a.config, seed_offset=rank * a.batch)
        extras = [config_pair(a.config, with_image=False, seed_offset=rank * a.batch + k) for k in range(1, a.batch)]
        res = Resident(pair, dev, a.batch, extras)
        units_solve = res.cells * a.batch * world
        units_warp = pair.final_w * pair.final_h * world
        scaling = "weak"

    for _ in range(a.warmup):
        res.solve(stream)
        res.warp(stream)
    run_solve, run_warp = (lambda: res.solve(stream)), (lambda: res.warp(stream))
    if a.graph and isinstance(res, Resident):
        # the entry points only enqueue kernels (no allocation, no synchronisation), so a step
        # can be captured once and replayed: one host call per step instead of one per kernel
        torch.cuda.synchronize()
        g_solve, g_warp = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_solve):
            res.solve(torch.cuda.current_stream().cuda_stream)
        with torch.cuda.graph(g_warp):
            res.warp(torch.cuda.current_stream().cuda_stream)
        run_solve, run_warp = g_solve.replay, g_warp.replay
        run_solve()
        run_warp()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run_solve()
    barrier()
    t_solve = time.perf_counter() - t0
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run_warp()
    barrier()
    t_warp = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([t_solve, t_warp], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_solve, t_warp = (float(v) for v in tt.cpu())
    assert int(res.status.cpu()[0]) == 0, "device status word set during the timed region"
    t_stitch = None
    if hasattr(res, "stitch"):      # extra: the fused stitch, same canvas, not part of `value`
        res.stitch(stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            res.stitch(stream)
        torch.cuda.synchronize()
        t_stitch = time.perf_counter() - t0
        res.warp(stream)            # leave the plain warped canvas in res.out for the byte count below
        torch.cuda.synchronize()

    t_eq = None
    if hasattr(res, "equalize"):    # extra: the pre-processing of both images is one call each
        res.equalize(stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            res.equalize(stream)
        torch.cuda.synchronize()
        t_eq = time.perf_counter() - t0

    t_ransac = None
    if hasattr(res, "ransac"):
        res.ransac(stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            res.ransac(stream)
        torch.cuda.synchronize()
        t_ransac = time.perf_counter() - t0

    # per-kernel durations, HIP events on the launch stream (rank-local)
    N.lib().apap_profile_enable(1)
    for _ in range(a.steps):
        res.solve(stream)
        res.warp(stream)
        if t_eq is not None:
            res.equalize(stream)
        if t_ransac is not None:
            res.ransac(stream)
    torch.cuda.synchronize()
    ms = (ctypes.c_float * N.PROF_SLOTS)()
    cnt = (ctypes.c_int * N.PROF_SLOTS)()
    N.check(N.lib().apap_profile_read(ms, cnt))
    N.lib().apap_profile_enable(0)
    kern = {k: (ms[i] / max(cnt[i], 1)) for i, k in enumerate(N.PROF_NAMES) if cnt[i] or i < 5}

    if rank == 0:
        local_cells = res.cells * getattr(res, "batch", 1)
        flops = K1_FLOPS_PER_CELL_POINT * res.n * local_cells
        t_k1 = kern["assemble"] * 1e-3
        achieved = flops / t_k1 / 1e12
        resolved = "valu" if a.variant == "valu" else "mfma"      # auto = mfma (apap_kernels.hip plan_solve)
        # HBM bytes per launch from the PMC counters, collected in separate rocprofv3 passes
        # (tools/profile.sh) and committed under profiles/: (2 x FETCH_SIZE + WRITE_SIZE) KiB,
        # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-B-per-lane reads.
        traffic, traffic_warp = None, None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile):
            tj = json.load(open(tfile))
            traffic = tj.get(f"{a.config}:k_assemble_{resolved}")
            traffic_warp = tj.get(f"{a.config}:k_warp_rows", tj.get(f"{a.config}:k_warp"))
        # K3 (HBM-bound half of the metric): 6 B per in-range pixel, 3 B per blank one
        out_pixels = pair.final_w * pair.final_h
        nz = int((res.out.view(-1, 3).amax(dim=1) > 0).sum().cpu()) if a.mode == "pairs" or world == 1 else None
        warp_bytes = (6 * nz + 3 * (out_pixels - nz)) if nz is not None else None
        line = {
            "metric": METRIC, "value": units_solve * a.steps / t_solve, "unit": "homographies/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": (t_solve + t_warp) / a.steps * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{a.config}: {CONFIGS[a.config][0]}x{CONFIGS[a.config][1]} pair, "
                                   f"{res.n} correspondences, {res.rows}x{res.cols} mesh, canvas "
                                   f"{pair.final_w}x{pair.final_h}", "mode": a.mode, "variant": a.variant,
                       "pairs_per_solve_launch": a.batch,
                       "launch": "hip graph replay" if a.graph else "eager",
                       "parallelism": f"{a.mode}x{world}"},
            "warp": {"value": units_warp * a.steps / t_warp / 1e6, "unit": "Mpix/s",
                     "ms_per_step": t_warp / a.steps * 1e3},
            "solve_ms_per_step": t_solve / a.steps * 1e3,
            "stitch": None if t_stitch is None else {
                "value": pair.final_w * pair.final_h * a.steps / t_stitch / 1e6, "unit": "Mpix/s (rank 0)",
                "ms_per_step": t_stitch / a.steps * 1e3,
                "note": "fused warp + paste + uniform_blend, apap.py:258-262; extra, not in `value`"},
            "equalize": None if t_eq is None else {
                "value": pair.shape[0] * pair.shape[1] * a.steps / t_eq / 1e6, "unit": "Mpix/s (rank 0)",
                "ms_per_step": t_eq / a.steps * 1e3,
                "roofline": {"kernels": "k_eq_hist + k_eq_apply", "bound": "hbm",
                             "achieved": 3.0 * pair.img.size / ((kern["eq_hist"] + kern["eq_apply"]) * 1e-3) / 1e9,
                             "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": 3.0 * pair.img.size / ((kern["eq_hist"] + kern["eq_apply"]) * 1e-3) / 1e9 / PEAK_HBM_GBS},
                "note": "per-channel cv.equalizeHist of the 4K source image, utils.py:85-91; algorithmic "
                        "traffic 3 bytes per image byte (read, read, write); extra, not in `value`"},
            "ransac": None if t_ransac is None else {
                "ms_per_call": t_ransac / a.steps * 1e3, "hypotheses": N.RANSAC_ITERATIONS, "points": res.n,
                "inliers": int(res.r_res.cpu()[1]),
                "note": "device half of the seed homography (4-point hypotheses, 5 px), baseline_stitch_test.py:42; "
                        "three small kernels, launch-latency-bound; extra, not in `value`"},
            "kernels_ms": kern,
            "roofline": {"kernel": "k_assemble_" + resolved, "bound": "mfma",
                         "achieved": achieved, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP64_TFLOPS, "traffic": traffic,
                         "hbm_gbs": None if traffic is None else traffic / t_k1 / 1e9,
                         "note": "fp64 FMA work: the f64 matrix and vector pipes share one 78.6 TF datasheet "
                                 "rate (54-59 TF sustained, profiles/r01_peak_fp64.txt); algorithmic 58 flop per "
                                 "(cell, keypoint) counts sqrt and exp as one flop each, the kernel executes "
                                 "~89 flops per pair for those 58 (DESIGN.md section 3)"},
            "roofline_warp": None if warp_bytes is None else {
                "kernel": "k_warp_rows", "bound": "hbm", "achieved": warp_bytes / (kern["warp"] * 1e-3) / 1e9,
                "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": warp_bytes / (kern["warp"] * 1e-3) / 1e9 / PEAK_HBM_GBS,
                "traffic": traffic_warp},
        }
        if not a.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(a.config, a.cpu_cells, a.cpu_rows,
                                                min(a.cpu_pool, os.cpu_count() or 1))
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
