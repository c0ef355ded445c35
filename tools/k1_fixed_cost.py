#!/usr/bin/env python3
"""The per-launch fixed cost of K1 / K2: solve times for very few keypoints (one or two LDS chunks) on the
C3 and C4 meshes, rocprof-free (HIP events of the context).   python tools/k1_fixed_cost.py"""
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
    for variant in (N.VARIANT_MFMA, N.VARIANT_VALU):
        ctx = N.Context(profile=1, variant=variant)
        for side, w, h in ((200, 3840, 2160), (400, 7680, 4320)):
            for n in (64, 128, 256, 512):
                p = synth_pair(w, h, n, side, seed=n + side, with_image=False)
                q = N.host_prepare(p.src, p.dst)
                t = torch.from_numpy(N.host_build_table(p.src, q["cf1"], q["cf2"])).to(dev)
                den = torch.from_numpy(N.host_build_denorm(q["iC2"], q["C1"], q["iN2"], q["N1"])).to(dev)
                vert = torch.from_numpy(np.ascontiguousarray(p.vertices.reshape(-1, 2))).to(dev)
                cells = vert.shape[0]
                H = torch.empty((cells, 9), dtype=torch.float32, device=dev)
                nbytes = max(N.lib().apap_solve_workspace_bytes(N._h(ctx), n, cells), 256)
                work = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                for rep in range(2):
                    for _ in range(20):
                        N.check(N.lib().apap_solve_device(N._h(ctx), t.data_ptr(), n, vert.data_ptr(), cells, 0.5, 100.0,
                                                          den.data_ptr(), H.data_ptr(), work.data_ptr(), nbytes, ctypes.c_void_p(0)))
                    torch.cuda.synchronize()
                    prof = ctx.profile_read()
                k1 = prof["assemble"][0] / prof["assemble"][1] * 1e3
                k2 = prof["eigen"][0] / max(prof["eigen"][1], 1) * 1e3
                print(f"variant {variant} {side}x{side} n={n:4d}: K1 {k1:7.1f} us  K2 {k2:6.1f} us", flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
