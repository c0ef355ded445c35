#!/usr/bin/env python3
"""What a host-buffer call of apap_local_homography is made of (C3): host set-up, the copies, the kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvx_proj_amd import _native as N
from cvx_proj_amd.dist import hip_solve
from cvx_proj_amd.synth import config_pair
p = config_pair(sys.argv[1] if len(sys.argv) > 1 else "C3", with_image=False)
def med(f, n=15):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e6
N.local_homography(p.src, p.dst, p.vertices, p.gamma, p.sigma, want_weights=False)
print("whole call            %.0f us" % med(lambda: N.local_homography(p.src, p.dst, p.vertices, p.gamma, p.sigma, want_weights=False)))
def host():
    q = N.host_prepare(p.src, p.dst)
    t = N.host_build_table(p.src, q["cf1"], q["cf2"]); d = N.host_build_denorm(q["iC2"], q["C1"], q["iN2"], q["N1"])
    return t, d
print("host set-up (ctypes)  %.0f us" % med(host))
t, d = host()
dev = torch.device("cuda", 0)
tt, dd = torch.from_numpy(t).to(dev), torch.from_numpy(d).to(dev)
vv = torch.from_numpy(np.ascontiguousarray(p.vertices.reshape(-1, 2))).to(dev)
def solve():
    hip_solve(tt, dd, vv, p.gamma, p.sigma); torch.cuda.synchronize()
solve()
print("resident solve + sync %.0f us" % med(solve))
H = hip_solve(tt, dd, vv, p.gamma, p.sigma)
print("H2D table (pageable)  %.0f us" % med(lambda: (tt.copy_(torch.from_numpy(t)), torch.cuda.synchronize())))
v_host = torch.from_numpy(np.ascontiguousarray(p.vertices.reshape(-1, 2)))
print("H2D vertices          %.0f us" % med(lambda: (vv.copy_(v_host), torch.cuda.synchronize())))
h_host = torch.empty(H.shape, dtype=torch.float32)
print("D2H H grid            %.0f us" % med(lambda: (h_host.copy_(H), torch.cuda.synchronize())))
pin = torch.empty(H.shape, dtype=torch.float32).pin_memory()
print("D2H H grid (pinned)   %.0f us" % med(lambda: (pin.copy_(H, non_blocking=True), torch.cuda.synchronize())))
pv = v_host.clone().pin_memory()
print("H2D vertices (pinned) %.0f us" % med(lambda: (vv.copy_(pv, non_blocking=True), torch.cuda.synchronize())))
