"""Study (GPU): the engine's grids of `cfg` pairs with the given seed offsets, saved for a value-by-value comparison with the
reference's grids on the CPU side (tests/studies/grid_mismatch.py).  With a golden name, only the pairs whose SHA-256 differs
from that fixture's are kept.   python tests/studies/engine_grids.py C3 out.npz 2 8      |      ... C4 out.npz --golden c4_seeds_sha"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cvx_proj_amd import _native as N  # noqa: E402
from cvx_proj_amd.synth import config_pair  # noqa: E402

cfg, out = sys.argv[1], sys.argv[2]
if sys.argv[3] == "--golden":
    g = np.load(os.path.join(ROOT, "tests", "golden", sys.argv[4] + ".npz"))
    seeds = [int(v) for v in (g["seeds"] if "seeds" in g.files else range(g["H_sha256"].shape[0]))]
    want = {k: g["H_sha256"][r].tobytes() for r, k in enumerate(seeds)}
else:
    seeds, want = [int(v) for v in sys.argv[3:]], None
grids, kept = [], []
for k in seeds:
    p = config_pair(cfg, with_image=False, seed_offset=k)
    H = N.local_homography(p.src, p.dst, p.vertices, p.gamma, p.sigma, want_weights=False)[0]
    if want is None or hashlib.sha256(H.tobytes()).digest() != want[k]:
        grids.append(H)
        kept.append(k)
np.savez_compressed(out, grids=np.stack(grids) if grids else np.zeros((0,)), bad=np.array(kept), cfg=cfg)
print("saved", kept, "of", cfg)
