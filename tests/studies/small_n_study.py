#!/usr/bin/env python3
"""How far do normal-equation solvers (the GPU engine, and numpy eigh) stray from the reference's
SVD of the weighted 2n x 9 system when it is barely determined (n = 5..12)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import apap_oracle as O
from cvx_proj_amd import _native as N
rng = np.random.default_rng(0)
rows = []
for n in (5, 6, 7, 8, 10, 12, 16, 32, 64):
    dg, df = [], []
    for trial in range(40):
        src = (rng.random((n, 2)) * [320, 240]).astype(np.float32)
        dst = (src @ np.array([[1.02, -0.03], [0.02, 0.97]], np.float32) + np.float32([5, -3])
               + rng.normal(0, 0.5, (n, 2)).astype(np.float32)).astype(np.float32)
        verts = rng.random((2, 2, 2)) * [320, 240]
        sigma = float(rng.choice([5.0, 30.0, 300.0]))
        Hl, _ = O.local_homography_loop(src, dst, verts, 0.3, sigma, want_weights=False)
        Hf, _ = O.local_homography_fast(src, dst, verts, 0.3, sigma)
        Hg, _ = N.local_homography(src, dst, verts, 0.3, sigma, want_weights=False)
        dg.append(O.reprojection_rmse_delta(Hg, Hl, src).max())
        df.append(O.reprojection_rmse_delta(Hf, Hl, src).max())
    dg, df = np.array(dg), np.array(df)
    print(f"n={n:3d}  GPU vs SVD: median {np.median(dg):.1e} max {dg.max():.1e} (>1e-4: {int((dg>1e-4).sum())}/40)   "
          f"eigh vs SVD: median {np.median(df):.1e} max {df.max():.1e} (>1e-4: {int((df>1e-4).sum())}/40)")
