#!/usr/bin/env python3
"""K3 forms side by side on one box: per-launch time of the gather kernel alone (tables kept: APAP_WARP_GATHER only), measured
two ways - HIP events around every single launch (the context's profiling slots: includes ~2 us of event overhead) and K
launches back to back between two events (what rocprofv3's kernel duration + the ~1 us launch gap add up to) - warm
(one image / canvas) and cold (a rotation of image / canvas sets larger than the Infinity Cache).

    python tools/warp_forms.py [--config C3] [--steps 40] [--forms "strips;strips,warp_rows=8;strips,warp_fast=0"]
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
    ap.add_argument("--config", default="C3")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--cold-mb", type=float, default=640.0)
    ap.add_argument("--forms", default="strips;strips,warp_rows=2;strips,warp_rows=8;strips,warp_fast=0")
    ap.add_argument("--stitch", action="store_true")
    ap.add_argument("--condition-s", type=float, default=0.25)
    ap.add_argument("--rows", type=int, default=0, help="warp only the first N canvas rows (a band): how the time scales with the number of strips")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    p = config_pair(a.config)
    rows, cols = p.vertices.shape[:2]
    H0, _ = N.local_homography(p.src, p.dst, p.vertices, p.gamma, p.sigma, want_weights=False)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
    mw, mh, H = t(p.mesh[0]), t(p.mesh[1]), t(H0.reshape(-1, 9))
    per = p.img.size + p.final_w * p.final_h * 3
    nsets = max(2, int(-(-a.cold_mb * 1e6 // per)))
    imgs = [t(p.img) for _ in range(nsets)]
    outs = [torch.zeros((p.final_h, p.final_w, 3), dtype=torch.uint8, device=dev) for _ in range(nsets)]
    center = torch.randint(0, 256, p.shape, dtype=torch.uint8, device=dev) if a.stitch else None
    wb = N.lib().apap_warp_workspace_bytes(rows, cols, p.final_w, p.final_h)
    work = torch.empty(wb, dtype=torch.uint8, device=dev)
    st = torch.zeros(1, dtype=torch.int32, device=dev)
    ref = None
    for form in a.forms.split(";"):
        opts = {}
        for tok in form.split(","):
            if tok == "strips":
                continue
            else:
                k, v = tok.split("=")
                opts[k] = int(v)
        ctx = N.Context(**opts)

        def launch(i, phases):
            N.check(N.lib().apap_warp_batch_device(ctx.handle, imgs[i].data_ptr(), 0, p.shape[0], p.shape[1],
                                                   None if center is None else center.data_ptr(), 0, p.shape[0], p.shape[1],
                                                   H.data_ptr(), rows, cols, mw.data_ptr(), mw.numel(), mh.data_ptr(), mh.numel(),
                                                   p.final_w, p.final_h, p.off_x, p.off_y, 0, a.rows or p.final_h, outs[i].data_ptr(), 0, None, 1,
                                                   phases, work.data_ptr(), wb, st.data_ptr(), ctypes.c_void_p(stream)))
        launch(0, N.WARP_ALL)
        torch.cuda.synchronize()
        if ref is None:
            ref = outs[0].clone()
        timing_only = os.environ.get("APAP_AB_TIMING_ONLY") == "1"      # a build under test that gives wrong pixels on purpose (ablations)
        assert timing_only or torch.equal(outs[0], ref), f"{form}: canvas differs from the first form's"
        res = {"config": a.config, "form": form, "rows": a.rows or p.final_h}
        for mode, n in (("warm", 1), ("cold", nsets)):
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < a.condition_s:       # sustained clocks
                for i in range(20):
                    launch(i % n, N.WARP_GATHER)
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(a.steps):
                launch(i % n, N.WARP_GATHER)
            e1.record()
            torch.cuda.synchronize()
            res[f"{mode}_back_to_back_us"] = round(e0.elapsed_time(e1) / a.steps * 1e3, 2)
            ctx.set("profile", 1)
            for i in range(a.steps):
                launch(i % n, N.WARP_GATHER)
            torch.cuda.synchronize()
            res[f"{mode}_events_per_launch_us"] = round(ctx.profile_read()["warp"][0] / a.steps * 1e3, 2)
            ctx.set("profile", 0)
            # the whole step (set-up + gather)
            t0 = time.perf_counter()
            for i in range(a.steps):
                launch(i % n, N.WARP_ALL)
            torch.cuda.synchronize()
            res[f"{mode}_step_with_setup_us"] = round((time.perf_counter() - t0) / a.steps * 1e6, 2)
        assert timing_only or int(st.cpu()[0]) == 0
        print(json.dumps(res), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
