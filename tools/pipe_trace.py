import sys, time
sys.path.insert(0, '.')
import numpy as np
from cvx_proj_amd import _native as N
from cvx_proj_amd.synth import config_pair
p = config_pair('C3')
H, _ = N.local_homography(p.src, p.dst, p.vertices, p.gamma, p.sigma, want_weights=False)
for i in range(4):
    t0 = time.perf_counter()
    N.local_warp(p.img, H, p.mesh[0], p.mesh[1], p.final_w, p.final_h, p.off_x, p.off_y)
    print('call', i, (time.perf_counter() - t0) * 1e3, 'ms', file=sys.stderr)
