#!/usr/bin/env python3
"""The multi-GPU expectation, so that a future hardware curve can be checked against it: for 2 / 4 / 8
ranks of one node, what ONE rank pays for its share - measured here on ONE GPU playing a middle rank, with the round's final
library - plus a MODEL of the collectives (they have never run over xGMI with N > 1 ranks):

* `cells` (BASELINE config 4: one 8K pair, 5000 keypoints, 400 x 400 mesh): the rank's mesh rows solved in the form
  ShardedSolver takes at that world size (two launches + the first all-gather beside the second for 2-4 ranks, one launch
  above), the all-gather of the H grid, the rank's own canvas band;
* `pairs` (config 5: 64 independent 4K pairs at 100 x 100): the rank's 64 / N pairs in one batched solve and one batched warp,
  no collective.

Collective model: an all-gather of B bytes per rank over N ranks of a fully connected xGMI node moves B bytes per link (every
rank sends its shard to each peer over that peer's own link): t = t0 + B / 153 GB/s, t0 = 25 us of RCCL launch + synchronisation
latency (a guess until measured).      python tools/scaling_model.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from cvx_proj_amd import _native as N  # noqa: E402
from cvx_proj_amd.dist import WarpPlan, hip_solve, hip_solve_batch, hip_warp_batch, row_partition  # noqa: E402
from cvx_proj_amd.synth import config_pair  # noqa: E402

LINK_GBS, T0_US = 153.0, 25.0
dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731


def bench(fn, reps, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def cells():
    p = config_pair("C4")
    rows, cols = p.vertices.shape[:2]
    q = N.host_prepare(p.src, p.dst)
    table, den = t(N.host_build_table(p.src, q["cf1"], q["cf2"])), t(N.host_build_denorm(q["iC2"], q["C1"], q["iN2"], q["N1"]))
    verts = p.vertices.reshape(-1, 2)
    work = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
    H_all = hip_solve(table, den, t(verts), p.gamma, p.sigma, work=work)
    img = t(p.img)
    edges = p.mesh[1]
    first = lambda k: int(np.ceil(edges[k])) if k < rows else p.final_h  # noqa: E731
    base = None
    print("cells (C4: 7680x4320, 5000 keypoints, 400x400 mesh, canvas %dx%d)" % (p.final_w, p.final_h))
    print("ranks | rows | solve shard us (form) | H all-gather: bytes per rank, model us | band rows, warp us | step us = solve + max(gather, warp) | H/s | eff.")
    for world in (1, 2, 4, 8):
        ra, rb = row_partition(rows, world)[min(world // 2, world - 1)]
        two = 2 <= world <= 4
        cuts = [(ra, ra + (rb - ra + 1) // 2), (ra + (rb - ra + 1) // 2, rb)] if two else [(ra, rb)]
        vs = [t(verts[a * cols:b * cols]) for a, b in cuts]
        outs = [torch.empty((v.shape[0], 9), dtype=torch.float32, device=dev) for v in vs]
        ctx = N.Context(plan_cells=0)
        t_solve = bench(lambda: [hip_solve(table, den, v, p.gamma, p.sigma, out=o, work=work, ctx=ctx) for v, o in zip(vs, outs)], 100, 100)
        ctx.close()
        shard_bytes = (rb - ra) * cols * 36
        t_gather = 0.0 if world == 1 else T0_US + shard_bytes / (LINK_GBS * 1e3)
        if two:     # the first half's gather runs beside the second launch: only the second half's gather is exposed
            t_gather = T0_US + (shard_bytes / 2) / (LINK_GBS * 1e3)
        # the rank's own band from its own rows (resident form: per-cell tables left by the solve's tail, gather kernel alone)
        own = edges[ra:rb + 1].copy()
        own[-1] = np.inf
        y0, y1 = first(ra), first(rb)
        plan = WarpPlan((p.mesh[0], own), (rb - ra, cols), p.final_w, p.final_h, p.off_x, p.off_y, dev)
        plan.cells(H_all[ra * cols:rb * cols])
        band = torch.empty((1, y1 - y0, p.final_w, 3), dtype=torch.uint8, device=dev)
        t_warp = bench(lambda: plan.gather(img, out=band, rows=(y0, y1 - y0)), 100, 30)
        assert plan.status_word() == 0
        step = t_solve + max(t_gather, t_warp)
        hs = rows * cols / (step * 1e-6)
        base = base or hs
        print(f"{world:5d} | {rb - ra:4d} | {t_solve:8.1f} ({'2 launches' if two else '1 launch'}) | {shard_bytes:9d} B, {t_gather:6.1f} | "
              f"{y1 - y0:5d}, {t_warp:6.1f} | {step:8.1f} | {hs:.3e} | {hs / base / world:.2f}", flush=True)
        del plan, band


def pairs():
    total = 64
    print("pairs (C5: 64 independent 3840x2160 pairs, 2000 keypoints, 100x100 mesh each; no collective)")
    print("ranks | pairs per rank | batched solve us | batched warp us (set-up + gather) | H/s | Mpix/s | eff. (solve)")
    loaded = [config_pair("C5", seed_offset=k) for k in range(8)]
    p0 = loaded[0]
    rows, cols = p0.vertices.shape[:2]
    vert = t(p0.vertices.reshape(-1, 2))
    mw, mh = t(p0.mesh[0]), t(p0.mesh[1])
    base = None
    for world in (1, 2, 4, 8):
        share = total // world
        # the share's keypoint sets and images: 8 distinct pairs cycled (the kernels' time does not depend on the values)
        prs = [loaded[k % 8] for k in range(share)]
        tabs, dens = [], []
        for pr in prs[:8]:
            q = N.host_prepare(pr.src, pr.dst)
            tabs.append(N.host_build_table(pr.src, q["cf1"], q["cf2"]))
            dens.append(N.host_build_denorm(q["iC2"], q["C1"], q["iN2"], q["N1"]))
        tables = t(np.stack([tabs[k % 8] for k in range(share)]))
        denorms = t(np.stack([dens[k % 8] for k in range(share)]))
        t_solve = bench(lambda: hip_solve_batch(tables, denorms, vert, p0.gamma, p0.sigma), 30, 10)
        H = hip_solve_batch(tables, denorms, vert, p0.gamma, p0.sigma)
        imgs = torch.stack([t(prs[k % 8].img) for k in range(min(share, 8))])
        imgs = imgs.repeat((share + 7) // 8, 1, 1, 1)[:share].contiguous()
        out = torch.empty((share, p0.final_h, p0.final_w, 3), dtype=torch.uint8, device=dev)
        t_warp = bench(lambda: hip_warp_batch(imgs, H, mw, mh, p0.final_w, p0.final_h, p0.off_x, p0.off_y, (rows, cols), out=out), 10, 3)
        hs = total * rows * cols / (t_solve * 1e-6)
        mp = total * p0.final_w * p0.final_h / (t_warp * 1e-6) / 1e6
        base = base or hs
        print(f"{world:5d} | {share:3d} | {t_solve:9.1f} | {t_warp:9.1f} | {hs:.3e} | {mp:.3e} | {hs / base / world:.2f}", flush=True)
        del tables, denorms, imgs, out, H


if __name__ == "__main__":
    print(f"# model, unmeasured on multi-GPU hardware: one MI355X playing one rank of N; collectives modelled at {LINK_GBS:.0f} GB/s per xGMI "
          f"link + {T0_US:.0f} us per all-gather")
    cells()
    pairs()
