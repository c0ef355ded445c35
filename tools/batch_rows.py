#!/usr/bin/env python3
"""K3 per pair against the strip height (APAP_OPT_WARP_ROWS = 4 / 5 / 6 / 8) for a batch of pairs in one launch (WarpPlan.gather,
launches back to back).  profiles/r05_k3_experiments.txt item 7.      python tools/batch_rows.py C5 32"""
import sys, os, json, ctypes, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cvx_proj_amd import _native as N
from cvx_proj_amd.dist import WarpPlan
from cvx_proj_amd.synth import config_pair
cfg, nb = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda", 0)
pairs = [config_pair(cfg, seed_offset=k) for k in range(min(nb, 4))]
p = pairs[0]
rows, cols = p.vertices.shape[:2]
H0, _ = N.local_homography(p.src, p.dst, p.vertices, p.gamma, p.sigma, want_weights=False)
H = torch.from_numpy(np.stack([H0.reshape(-1, 9)] * nb)).to(dev).view(-1, 9)
imgs = torch.stack([torch.from_numpy(pairs[k % len(pairs)].img) for k in range(nb)]).to(dev)
for r in (4, 5, 6, 8):
    ctx = N.Context(warp_rows=r)
    plan = WarpPlan(p.mesh, (rows, cols), p.final_w, p.final_h, p.off_x, p.off_y, dev, batch=nb, ctx=ctx)
    plan.cells(H)
    out = plan.gather(imgs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        plan.gather(imgs, out=out)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        plan.gather(imgs, out=out)
    e1.record(); torch.cuda.synchronize()
    print(f"{cfg} x{nb} rows {r}: {e0.elapsed_time(e1) / 20 / nb * 1e3:.2f} us per pair", flush=True)
    ctx.close()
