#!/usr/bin/env python3
"""Host-side timeline of apap_local_warp's overlapped path: build the library with -DAPAP_TRACE_PIPE into
tools/variants/lib_trace_pipe.so, then  APAP_HIP_LIB=$PWD/tools/variants/lib_trace_pipe.so python tools/pipe_trace.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from cvx_proj_amd import _native as N  # noqa: E402
from cvx_proj_amd.synth import config_pair  # noqa: E402

p = config_pair("C3")
H, _ = N.local_homography(p.src, p.dst, p.vertices, p.gamma, p.sigma, want_weights=False)


CTX = N.Context(overlap_pcie=1)


def calls(tag, n=3, img=None, out=None, ctx=CTX):
    for i in range(n):
        t0 = time.perf_counter()
        N.local_warp(p.img if img is None else img, H, p.mesh[0], p.mesh[1], p.final_w, p.final_h, p.off_x, p.off_y, out=out, ctx=ctx)
        print(tag, i, f"{(time.perf_counter() - t0) * 1e3:.3f} ms", file=sys.stderr)


calls("fresh process")
N.local_homography(p.src, p.dst, p.vertices, p.gamma, p.sigma, want_weights=True)     # 640 MB through the host allocator
calls("after a 640 MB array came and went")
big = [np.ones(30_000_000, np.uint8) for _ in range(4)]
del big
calls("after four 30 MB arrays came and went")

import torch  # noqa: E402
pin_img = torch.empty(p.img.shape, dtype=torch.uint8, pin_memory=True)
pin_img.numpy()[...] = p.img
pin_out = torch.empty((p.final_h, p.final_w, 3), dtype=torch.uint8, pin_memory=True)
ref, _ = N.local_warp(p.img, H, p.mesh[0], p.mesh[1], p.final_w, p.final_h, p.off_x, p.off_y)
calls("image and canvas in page-locked memory of the caller", 6, pin_img.numpy(), pin_out.numpy())
calls("the same, APAP_OPT_OVERLAP_PCIE off (one copy up, the kernels, one copy down)", 4, pin_img.numpy(), pin_out.numpy(), ctx=None)
print("same canvas:", bool((pin_out.numpy() == ref).all()), file=sys.stderr)
