"""Study (GPU): the 64 grids of config 5 from the engine, their SHA-256 against tests/golden/c5_all_sha.npz, saved for a value-by-value
comparison with the reference's grids on the CPU side.   python tests/studies/c5_all_grids.py out.npz"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cvx_proj_amd.dist import solve_pairs  # noqa: E402
from cvx_proj_amd.synth import config_pair  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "c5_all_sha.npz"))
dev = torch.device("cuda:0")
grids, bad = [], []
for lo in range(0, 64, 16):
    pairs = [config_pair("C5", with_image=False, seed_offset=k) for k in range(lo, lo + 16)]
    for i, H in enumerate(solve_pairs(pairs, dev)):
        grids.append(H)
        if hashlib.sha256(H.tobytes()).digest() != g["H_sha256"][lo + i].tobytes():
            bad.append(lo + i)
print("pairs whose grid differs from the reference's:", bad)
np.savez_compressed(sys.argv[1], grids=np.stack(grids), bad=np.array(bad))
