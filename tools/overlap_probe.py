#!/usr/bin/env python3
"""Can the arithmetic and the byte movement of K3 overlap at all on this chip?  Two experiment builds of the library (wrong
pixels, timing only) - one whose warp kernel does all table / record loads and arithmetic but touches no image
(-DAPAP_K3_ABL_NOGATHER -DAPAP_K3_ABL_NOSTORE), one that does no coordinate work and only moves the bytes (-DAPAP_K3_ABL_COPY)
- are loaded side by side and run alone and CONCURRENTLY on two streams.
    tools/overlap_probe.py tools/variants/lib_k3_ARITH.so tools/variants/lib_k3_COPY.so"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from cvx_proj_amd import _native as N  # noqa: E402
from cvx_proj_amd.synth import config_pair  # noqa: E402

p = config_pair("C3")
dev = torch.device("cuda", 0)
H, _ = N.local_homography(p.src, p.dst, p.vertices, p.gamma, p.sigma, want_weights=False)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
img, Hd, mw, mh = t(p.img), t(H.reshape(-1, 9)), t(p.mesh[0]), t(p.mesh[1])
rows, cols = H.shape[:2]


class Side:
    def __init__(self, path):
        self.lib = C.CDLL(path)
        self.lib.apap_warp_workspace_bytes.restype = C.c_size_t
        self.lib.apap_warp_device.argtypes = N.SIGNATURES["apap_warp_device"][1]
        nb = self.lib.apap_warp_workspace_bytes(rows, cols, p.final_w, p.final_h)
        self.nb = nb
        self.work = torch.empty(nb, dtype=torch.uint8, device=dev)
        self.out = torch.zeros((p.final_h, p.final_w, 3), dtype=torch.uint8, device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self.stream = torch.cuda.Stream(dev)

    def launch(self):
        rc = self.lib.apap_warp_device(None, img.data_ptr(), p.shape[0], p.shape[1], Hd.data_ptr(), rows, cols, mw.data_ptr(),
                                       p.mesh.shape[1], mh.data_ptr(), p.mesh.shape[1], p.final_w, p.final_h, p.off_x, p.off_y,
                                       self.out.data_ptr(), None, self.work.data_ptr(), self.nb, self.status.data_ptr(),
                                       C.c_void_p(self.stream.cuda_stream))
        assert rc == 0


def timed(sides, n=300):
    for _ in range(50):
        for s in sides:
            s.launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for s in sides:
            s.launch()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


a, b = Side(sys.argv[1]), Side(sys.argv[2])
ta, tb, tab = timed([a]), timed([b]), timed([a, b])
print(f"set-up + warp kernel per step: {os.path.basename(sys.argv[1])} alone {ta:.1f} us, {os.path.basename(sys.argv[2])} alone {tb:.1f} us, "
      f"both on two streams {tab:.1f} us (sum {ta + tb:.1f}, max {max(ta, tb):.1f})")
