"""Study (CPU, imports the reference and the oracle): the C5 pairs whose engine grid does not hash to the reference's -
which float32 values differ, by how much, and what a 60-digit SVD of the reference's own matrix says about them.
    python tests/studies/c5_mismatch.py gpurun_out/r6h/c5_grids.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as MG  # noqa: E402
from oracle import apap_oracle as O  # noqa: E402
from cvx_proj_amd.synth import config_pair  # noqa: E402

z = np.load(sys.argv[1])
ref_apap, ref_utils = MG.import_reference()
for k in z["bad"]:
    p = config_pair("C5", with_image=False, seed_offset=int(k))
    eng = ref_apap.APAP(p.gamma, p.sigma, [p.final_w, p.final_h], [p.off_x, p.off_y])
    H_ref, _ = eng.local_homography(p.src, p.dst, p.vertices)
    H_gpu = z["grids"][int(k)]
    diff = np.argwhere(H_ref != H_gpu)
    ulp = np.abs(H_ref.view(np.int32).astype(np.int64) - H_gpu.view(np.int32).astype(np.int64))
    d = O.reprojection_rmse_delta(H_gpu, H_ref, p.src)
    print(f"pair {k}: {len(diff)} of {H_ref.size} float32 values differ, max {ulp.max()} ulp, rmse delta max {d.max():.3e} px")
    H_fast, _ = O.local_homography_fast(p.src, p.dst, p.vertices, p.gamma, p.sigma)
    print(f"   oracle (eigh of the normal matrix) vs reference: {int((H_fast != H_ref).sum())} differ; vs engine: {int((H_fast != H_gpu).sum())} differ")
    for i, j, a, b in diff[:6]:
        exact = O.local_homography_exact_cell(p.src, p.dst, p.vertices[i, j], p.gamma, p.sigma)
        # the float64 value before rounding, from the exact path, against the two float32 candidates
        print(f"   cell ({i},{j}) entry ({a},{b}): reference {H_ref[i, j, a, b]!r}  engine {H_gpu[i, j, a, b]!r}  60-digit SVD {exact[a, b]!r}"
              f"  -> {'engine' if exact[a, b] == H_gpu[i, j, a, b] else 'reference' if exact[a, b] == H_ref[i, j, a, b] else 'neither'} has the exact answer's float32")
