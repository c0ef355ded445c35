#!/usr/bin/env python3
"""What ONE rank of 8 pays for its band of the C4 warp: set-up over the whole 400 x 400 grid (round 2: bands of equal height,
warped from the gathered grid) against set-up over its own 50 mesh rows (bands aligned to the mesh rows, cvx_proj_amd/dist.py)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from cvx_proj_amd import _native as N  # noqa: E402
from cvx_proj_amd.dist import hip_warp_rows, row_partition  # noqa: E402
from cvx_proj_amd.synth import config_pair  # noqa: E402

p = config_pair("C4")
dev = torch.device("cuda", 0)
H, _ = N.local_homography(p.src, p.dst, p.vertices, p.gamma, p.sigma, want_weights=False)
rows, cols = H.shape[:2]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
img, Hd, mw = t(p.img), t(H.reshape(-1, 9)), t(p.mesh[0])
world, rank = 8, 3
ra, rb = row_partition(rows, world)[rank]
edges = p.mesh[1]
first = lambda k: int(np.ceil(edges[k])) if k < rows else p.final_h  # noqa: E731
band_aligned = (first(ra), first(rb))
band_equal = row_partition(p.final_h, world)[rank]
own = edges[ra:rb + 1].copy()
own[-1] = np.inf


def run(label, Hview, mesh_h, shape, band):
    y0, y1 = band
    out = torch.zeros((y1 - y0, p.final_w, 3), dtype=torch.uint8, device=dev)
    nb = N.lib().apap_warp_workspace_bytes(shape[0], shape[1], p.final_w, p.final_h)
    kw = dict(work=torch.empty(nb, dtype=torch.uint8, device=dev), status=torch.zeros(1, dtype=torch.int32, device=dev))
    for _ in range(20):
        hip_warp_rows(img, Hview, mw, mesh_h, p.final_w, p.final_h, p.off_x, p.off_y, y0, y1 - y0, out, shape, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        hip_warp_rows(img, Hview, mw, mesh_h, p.final_w, p.final_h, p.off_x, p.off_y, y0, y1 - y0, out, shape, **kw)
    torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per band of {y1 - y0} rows")
    return out


a = run("whole grid, equal band ", Hd, t(edges), (rows, cols), band_equal)
b = run("own mesh rows, own band", Hd[ra * cols:rb * cols], t(own), (rb - ra, cols), band_aligned)
full, _ = N.local_warp(p.img, H, p.mesh[0], p.mesh[1], p.final_w, p.final_h, p.off_x, p.off_y)
assert np.array_equal(a.cpu().numpy(), full[band_equal[0]:band_equal[1]]) and np.array_equal(b.cpu().numpy(), full[band_aligned[0]:band_aligned[1]])
print("both bands equal the single-GPU canvas")
