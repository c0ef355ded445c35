#!/usr/bin/env python3
"""Warp rate per pair against the number of pairs per launch (apap_warp_batch_device, grid.z = pair), resident data,
every pair with its own source image and canvas.  One warp step of a single pair leaves the chip partly idle (a set-up
kernel of one wave per SIMD, then 1.45 generations of gather waves at C3; 0.2-0.4 at C1 / C2); batching fills it.

    python tools/batch_warp_rate.py [--configs C1,C2,C5,C3] [--batches 1,2,8,32] [--steps 30]
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvx_proj_amd import _native as N  # noqa: E402
from cvx_proj_amd.synth import config_pair  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="C1,C2,C5,C3")
    ap.add_argument("--batches", default="1,2,8,32")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--phases", type=int, default=N.WARP_ALL)
    a = ap.parse_args()
    ctx = N.Context()
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    for cfg in a.configs.split(","):
        p = config_pair(cfg)
        rows, cols = p.vertices.shape[:2]
        H0, _ = N.local_homography(p.src, p.dst, p.vertices, p.gamma, p.sigma, want_weights=False)
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
        mw, mh = t(p.mesh[0]), t(p.mesh[1])
        base = None
        for b in [int(x) for x in a.batches.split(",")]:
            imgs = t(p.img).unsqueeze(0).repeat(b, 1, 1, 1).contiguous()
            H = t(H0.reshape(1, -1, 9)).repeat(b, 1, 1).contiguous()
            out = torch.zeros((b, p.final_h, p.final_w, 3), dtype=torch.uint8, device=dev)
            wb = N.lib().apap_warp_batch_workspace_bytes(rows, cols, p.final_w, p.final_h, b)
            work = torch.empty(wb, dtype=torch.uint8, device=dev)
            st = torch.zeros(1, dtype=torch.int32, device=dev)

            def step(phases=a.phases):
                N.check(N.lib().apap_warp_batch_device(ctx.handle, imgs.data_ptr(), imgs[0].numel(), p.shape[0], p.shape[1], None, 0, 0, 0,
                                                       H.data_ptr(), rows, cols, mw.data_ptr(), mw.numel(), mh.data_ptr(), mh.numel(),
                                                       p.final_w, p.final_h, p.off_x, p.off_y, 0, p.final_h, out.data_ptr(),
                                                       out[0].numel(), None, b, phases, work.data_ptr(), wb, st.data_ptr(),
                                                       ctypes.c_void_p(stream)))
            step(N.WARP_ALL)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.2:       # sustained clocks
                for _ in range(10):
                    step()
                torch.cuda.synchronize()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / a.steps / b * 1e6
            assert int(st.cpu()[0]) == 0 and torch.equal(out[0], out[-1])
            base = base or us
            print(json.dumps({"config": cfg, "canvas": [p.final_w, p.final_h], "mesh": rows, "pairs_per_launch": b,
                              "phases": a.phases, "us_per_pair": round(us, 2), "mpix_per_s": round(p.final_w * p.final_h / us, 1),
                              "rate_vs_single_launch": round(base / us, 2)}), flush=True)
            del imgs, H, out, work


if __name__ == "__main__":
    main()
