#!/usr/bin/env python3
"""apap_local_warp with APAP_OPT_OVERLAP_PCIE = 1 on C3: upload chunks x download bands x order (small uploads + set-up before or after the image
chunks are enqueued), median host clock of 15 calls on reused buffers.  Ran against a diagnostic build of round 4 that read the three environment variables
(removed again: the library reads no environment; profiles/r04_pcie_duplex.txt has the table) - kept as the record of the method:  APAP_HIP_LIB=$PWD/tools/variants/lib_trace_pipe.so python tools/ovl_sweep.py 2>/dev/null"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from cvx_proj_amd import _native as N  # noqa: E402
from cvx_proj_amd.synth import config_pair  # noqa: E402

p = config_pair("C3")
H, _ = N.local_homography(p.src, p.dst, p.vertices, p.gamma, p.sigma, want_weights=False)
ref, _ = N.local_warp(p.img, H, p.mesh[0], p.mesh[1], p.final_w, p.final_h, p.off_x, p.off_y)
ovl = N.Context(overlap_pcie=1)
out = np.empty_like(ref)


def med(ctx, n=15, want_inverse=True):
    ts = []
    for i in range(n + 3):
        t0 = time.perf_counter()
        N.local_warp(p.img, H, p.mesh[0], p.mesh[1], p.final_w, p.final_h, p.off_x, p.off_y, ctx=ctx, out=out, want_inverse=want_inverse)
        ts.append(time.perf_counter() - t0)
    assert np.array_equal(out, ref)
    ts = sorted(ts[3:])
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3


print("sequential (default): median %.3f ms, min %.3f" % med(None))
for first in (0, 1):
    for chunks in (2, 4, 8):
        for bands in (2, 4, 8, 12):
            os.environ["APAP_PIPE_SMALL_FIRST"], os.environ["APAP_PIPE_CHUNKS"], os.environ["APAP_PIPE_BANDS"] = str(first), str(chunks), str(bands)
            m, lo = med(ovl)
            m2, _ = med(ovl, 9, False)
            print(f"small uploads first {first}  chunks {chunks:2d}  bands {bands:2d}: median {m:.3f} ms  min {lo:.3f}   without the inverses coming back {m2:.3f}", flush=True)
