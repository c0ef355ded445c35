#!/usr/bin/env python3
"""Where a wave of the strip kernel spends its life: per-wave wall-clock stamps (100 MHz) inside k_warp_fast<false, 4> from a
-DAPAP_K3_TRACE build (every stamp is preceded by s_waitcnt vmcnt(0) lgkmcnt(0): "everything issued so far has completed").

    hipcc ... -DAPAP_K3_TRACE -shared -o tools/variants/lib_k3trace.so <sources>
    APAP_HIP_LIB=$PWD/tools/variants/lib_k3trace.so python tools/k3_trace.py [C3]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvx_proj_amd import _native as N  # noqa: E402
from cvx_proj_amd.synth import config_pair  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
p = config_pair(cfg)
dev = torch.device("cuda:0")
rows, cols = p.vertices.shape[:2]
H0, _ = N.local_homography(p.src, p.dst, p.vertices, p.gamma, p.sigma, want_weights=False)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
mw, mh, H, img = t(p.mesh[0]), t(p.mesh[1]), t(H0.reshape(-1, 9)), t(p.img)
out = torch.zeros((p.final_h, p.final_w, 3), dtype=torch.uint8, device=dev)
wb = N.lib().apap_warp_workspace_bytes(rows, cols, p.final_w, p.final_h)
work = torch.empty(wb, dtype=torch.uint8, device=dev)
st = torch.zeros(1, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream().cuda_stream


def launch(phases):
    N.check(N.lib().apap_warp_batch_device(None, img.data_ptr(), 0, p.shape[0], p.shape[1], None, 0, 0, 0, H.data_ptr(), rows, cols,
                                           mw.data_ptr(), mw.numel(), mh.data_ptr(), mh.numel(), p.final_w, p.final_h, p.off_x, p.off_y, 0,
                                           p.final_h, out.data_ptr(), 0, None, 1, phases, work.data_ptr(), wb, st.data_ptr(),
                                           C.c_void_p(stream)))


launch(N.WARP_ALL)
for _ in range(200):
    launch(N.WARP_GATHER)
torch.cuda.synchronize()
launch(N.WARP_GATHER)       # the traced launch: the last one to write the stamps
torch.cuda.synchronize()
n = 16384
buf = np.zeros(n * 8, np.int64)
fn = N.lib().apap_debug_k3_trace
fn.argtypes = [C.c_void_p, C.c_int]
fn.restype = C.c_int
assert fn(buf.ctypes.data, n * 8) == 0
s = buf.reshape(n, 8)[:, :6].astype(np.float64)
live = (s[:, 0] > 0) & (s[:, 5] > s[:, 0])
s = s[live]
t0 = s[:, 0].min()
us = (s - t0) / 100.0           # 100 MHz wall clock -> microseconds
names = ["entry -> column / row entries", "-> records", "-> offsets computed", "-> gathers landed (+ exact path)", "-> stores acknowledged"]
print(f"{cfg}: {len(s)} waves traced; kernel span {us[:, 5].max():.2f} us (first entry to last exit)")
print(f"wave start times: median {np.median(us[:, 0]):.2f} us, 90 % {np.percentile(us[:, 0], 90):.2f}, max {us[:, 0].max():.2f}")
print(f"wave lifetimes:   median {np.median(us[:, 5] - us[:, 0]):.2f} us, 90 % {np.percentile(us[:, 5] - us[:, 0], 90):.2f}, max {(us[:, 5] - us[:, 0]).max():.2f}")
for k, name in enumerate(names):
    d = us[:, k + 1] - us[:, k]
    print(f"  {name:38s} median {np.median(d):6.2f} us   mean {d.mean():6.2f}   90 % {np.percentile(d, 90):6.2f}   max {d.max():6.2f}")
# by generation: waves that started in the first microsecond against the rest
first = us[:, 0] < 1.5
for tag, m in (("first generation (started < 1.5 us)", first), ("later waves", ~first)):
    if m.sum():
        d = us[m]
        print(f"{tag}: {int(m.sum())} waves, phases (median us): " + "  ".join(f"{np.median(d[:, k + 1] - d[:, k]):.2f}" for k in range(5))
              + f"   life {np.median(d[:, 5] - d[:, 0]):.2f}   exit {np.median(d[:, 5]):.2f}")
hist, edges = np.histogram(us[:, 5], bins=12)
print("exit-time histogram (us):", "  ".join(f"{edges[i]:.1f}-{edges[i + 1]:.1f}: {hist[i]}" for i in range(len(hist))))
