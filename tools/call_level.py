#!/usr/bin/env python3
"""Call-level (numpy in -> numpy out, PCIe-inclusive) rates of the host-buffer entry points on C3."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cvx_proj_amd import _native as N
from cvx_proj_amd.synth import config_pair
p = config_pair(sys.argv[1] if len(sys.argv) > 1 else "C3")
H, _ = N.local_homography(p.src, p.dst, p.vertices, p.gamma, p.sigma, want_weights=False)
def best(f, n=7):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return min(ts), sorted(ts)[len(ts) // 2]
cells = H.shape[0] * H.shape[1]
b, m = best(lambda: N.local_homography(p.src, p.dst, p.vertices, p.gamma, p.sigma, want_weights=False))
print(f"local_homography (no W): best {b*1e3:.2f} ms median {m*1e3:.2f} ms -> {cells/m:.3e} H/s call-level")
b, m = best(lambda: N.local_warp(p.img, H, p.mesh[0], p.mesh[1], p.final_w, p.final_h, p.off_x, p.off_y), 5)
print(f"local_warp: best {b*1e3:.2f} ms median {m*1e3:.2f} ms -> {p.final_w*p.final_h/m/1e6:.0f} Mpix/s call-level")
b, m = best(lambda: N.local_homography(p.src, p.dst, p.vertices, p.gamma, p.sigma, want_weights=True), 3)
print(f"local_homography (with the {cells*len(p.src)*8/1e6:.0f} MB W tensor): median {m*1e3:.1f} ms -> {cells/m:.3e} H/s")
