#!/usr/bin/env python3
"""apap_local_warp with APAP_OPT_OVERLAP_PCIE = 1, call by call (host clock); with a -DAPAP_TRACE_PIPE build in APAP_HIP_LIB the library
prints its own timeline of every call on stderr."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvx_proj_amd import _native as N  # noqa: E402
from cvx_proj_amd.synth import config_pair  # noqa: E402

p = config_pair("C3")
H, _ = N.local_homography(p.src, p.dst, p.vertices, p.gamma, p.sigma, want_weights=False)
ovl = N.Context(overlap_pcie=1)
for ctx, tag in ((None, "sequential"), (ovl, "overlapped")):
    for i in range(6):
        t0 = time.perf_counter()
        N.local_warp(p.img, H, p.mesh[0], p.mesh[1], p.final_w, p.final_h, p.off_x, p.off_y, ctx=ctx)
        print(tag, i, f"{(time.perf_counter() - t0) * 1e3:.3f} ms", file=sys.stderr, flush=True)
