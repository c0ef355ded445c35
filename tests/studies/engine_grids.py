"""Study (GPU): the engine's grids of `cfg` pairs with the given seed offsets, saved for a value-by-value comparison with the
reference's grids on the CPU side (tests/studies/grid_mismatch.py).   python tests/studies/engine_grids.py C3 out.npz 2 8"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cvx_proj_amd import _native as N  # noqa: E402
from cvx_proj_amd.synth import config_pair  # noqa: E402

cfg, out, seeds = sys.argv[1], sys.argv[2], [int(v) for v in sys.argv[3:]]
grids = []
for k in seeds:
    p = config_pair(cfg, with_image=False, seed_offset=k)
    grids.append(N.local_homography(p.src, p.dst, p.vertices, p.gamma, p.sigma, want_weights=False)[0])
np.savez_compressed(out, grids=np.stack(grids), bad=np.array(seeds), cfg=cfg)
print("saved", len(seeds), "grids of", cfg)
