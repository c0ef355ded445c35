"""The careful path as an algorithm (oracle/careful_oracle.py: Givens QR of the weighted rows + one-sided
Jacobi on R^T), checked on CPU against the golden vectors: the exact 60-digit answers where the
reference's own float64 SVD is lost, the reference's grids where it is not, and the edge cases with fewer
than 5 keypoints.  The GPU suite checks that the kernel computes the same thing."""
import os
import sys

import numpy as np
import pytest

from oracle import apap_oracle as O
from oracle import careful_oracle as K

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_gpu_fuzz import random_case  # noqa: E402  (pure numpy; its tests are gpu-marked, the generator is not)


@pytest.mark.parametrize("seed", [1974, 5634, 6012])
def test_spec_equals_the_exact_answer_where_the_reference_is_lost(golden, seed):
    g = golden("illcond_truth")
    c = random_case(1000 + seed)
    cells, exact, H_ref = g[f"cells{seed}"], g[f"exact{seed}"], g[f"H{seed}"]
    cols = c["verts"].shape[1]
    pick = np.random.default_rng(seed).choice(len(cells), size=min(60, len(cells)), replace=False)
    idx = [(int(cells[k] // cols), int(cells[k] % cols)) for k in pick]
    H = K.local_homography_careful(c["src"], c["dst"], c["verts"], c["gamma"], c["sigma"], cells=idx)
    mine = np.stack([H[i, j] for i, j in idx])
    d_exact = O.reprojection_rmse_delta(mine, exact[pick], c["src"])
    d_ref = O.reprojection_rmse_delta(np.stack([H_ref[i, j] for i, j in idx]), exact[pick], c["src"])
    assert d_exact.max() < 1e-4, d_exact.max()
    assert d_ref.max() > d_exact.max()          # the algorithm is closer to the exact answer than the reference


@pytest.mark.parametrize("seed", [544, 883])
def test_spec_vs_the_reference_on_round_1_soak_failures(golden, seed):
    g = golden("illcond_ref")
    c = random_case(1000 + seed)
    rows, cols = c["verts"].shape[:2]
    rng = np.random.default_rng(seed)
    idx = [(int(i), int(j)) for i, j in zip(rng.integers(0, rows, 40), rng.integers(0, cols, 40))]
    H = K.local_homography_careful(c["src"], c["dst"], c["verts"], c["gamma"], c["sigma"], cells=idx)
    d = O.reprojection_rmse_delta(np.stack([H[i, j] for i, j in idx]), np.stack([g[f"H{seed}"][i, j] for i, j in idx]), c["src"])
    assert d.max() < 1e-4


def test_spec_on_well_conditioned_and_short_systems(golden):
    """Bit-identical float32 grids on the tiny case and on the edge cases with 4, 5 and 6 keypoints (n = 4: the
    thin SVD keeps 8 vectors, V[-1] is not the null vector)."""
    g = golden("tiny_sigma6")
    H = K.local_homography_careful(g["src"], g["dst"], g["vertices"], float(g["gamma"]), float(g["sigma"]))
    assert np.array_equal(H, g["H_ref"])
    e = golden("edge_ref")
    for k in (0, 1, 2):
        gamma, sigma = (float(v) for v in e[f"par{k}"])
        H = K.local_homography_careful(e[f"src{k}"], e[f"dst{k}"], e[f"verts{k}"], gamma, sigma)
        assert np.array_equal(H, e[f"H{k}"]), k


def test_exact_cell_arbiter_reproduces_the_committed_exact_answers(golden):
    """oracle.local_homography_exact_cell (the arbiter of the GPU fuzz test) against the 60-digit answers that
    tests/golden/make_golden.py stored: same float32 matrices."""
    pytest.importorskip("mpmath")
    g = golden("illcond_truth")
    seed = 5634
    c = random_case(1000 + seed)
    cells, exact = g[f"cells{seed}"], g[f"exact{seed}"]
    cols = c["verts"].shape[1]
    for k in (0, len(cells) // 2, len(cells) - 1):
        i, j = int(cells[k] // cols), int(cells[k] % cols)
        mine = O.local_homography_exact_cell(c["src"], c["dst"], c["verts"][i, j], c["gamma"], c["sigma"])
        assert np.array_equal(mine, exact[k]), (k, mine, exact[k])
