#!/usr/bin/env python3
"""K1 / K2 kernel times over a (cells, keypoints) grid: where does the per-pair cost of the assemble
kernel depend on the mesh size, where on the keypoint count?   python tools/k1_scan.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from cvx_proj_amd import _native as N
from cvx_proj_amd.synth import synth_pair


def main():
    dev = torch.device("cuda:0")
    ctx = N.Context(profile=1)
    for side, w, h in ((100, 1920, 1080), (200, 3840, 2160), (283, 5430, 3054), (400, 7680, 4320)):
        for n in (500, 1000, 2000, 5000):
            p = synth_pair(w, h, n, side, seed=n + side, with_image=False)
            q = N.host_prepare(p.src, p.dst)
            t = torch.from_numpy(N.host_build_table(p.src, q["cf1"], q["cf2"])).to(dev)
            den = torch.from_numpy(N.host_build_denorm(q["iC2"], q["C1"], q["iN2"], q["N1"])).to(dev)
            vert = torch.from_numpy(np.ascontiguousarray(p.vertices.reshape(-1, 2))).to(dev)
            cells = vert.shape[0]
            H = torch.empty((cells, 9), dtype=torch.float32, device=dev)
            nbytes = max(N.lib().apap_solve_workspace_bytes(N._h(ctx), n, cells), 256)
            work = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            for _ in range(12):
                N.check(N.lib().apap_solve_device(N._h(ctx), t.data_ptr(), n, vert.data_ptr(), cells, 0.5, 100.0, den.data_ptr(),
                                                  H.data_ptr(), work.data_ptr(), nbytes, ctypes.c_void_p(0)))
            torch.cuda.synchronize()
            prof = ctx.profile_read()
            k1 = prof["assemble"][0] / prof["assemble"][1] * 1e3
            k2 = prof["eigen"][0] / max(prof["eigen"][1], 1) * 1e3
            pairs = cells * n
            cyc = k1 * 1e-6 * 2.3e9 * 1024 / (pairs / 64)
            print(f"{side}x{side} cells, n={n:5d}: K1 {k1:8.1f} us  K2 {k2:6.1f} us  "
                  f"~{cyc:5.0f} issue cycles per 64 pairs (at 2.3 GHz)", flush=True)


if __name__ == "__main__":
    main()
