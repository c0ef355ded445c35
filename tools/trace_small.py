#!/usr/bin/env python3
"""Where the fused small-mesh launch (k_solve_small, config C1) spends its time: run against a library built
with -DAPAP_TRACE_SMALL, whose kernel prints s_memtime stamps of block 3 (prologue / step loop / reduction /
tail, and the tail's factorisation / solves / guard / de-normalisation).
    hipcc ... -DAPAP_TRACE_SMALL -shared -o tools/variants/lib_trace.so cvx_proj_amd/csrc/*.cpp cvx_proj_amd/csrc/*.hip
    APAP_HIP_LIB=$PWD/tools/variants/lib_trace.so python tools/trace_small.py"""
import sys
sys.path.insert(0, '.')
import numpy as np
from cvx_proj_amd import _native as N
from cvx_proj_amd.synth import config_pair
p = config_pair("C1", with_image=False)
for _ in range(4):
    H, _ = N.local_homography(p.src, p.dst, p.vertices, p.gamma, p.sigma, want_weights=False)
print(H.shape)
