"""Property tests (hypothesis) of the host half and of the oracle's two flavours.  CPU only."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import apap_oracle as O


@settings(max_examples=40, deadline=None)
@given(n=st.integers(2, 9000), scale=st.sampled_from([1.0, 50.0, 4000.0, 1e5]), seed=st.integers(0, 2**31 - 1),
       degenerate=st.sampled_from(["none", "same_x", "same_point"]))
def test_c_setup_is_bitwise_numpy(native, n, scale, seed, degenerate):
    rng = np.random.default_rng(seed)
    src = (rng.random((n, 2)) * scale).astype(np.float32)
    dst = (src * np.float32(1.01) + rng.normal(0, 0.01 * scale, (n, 2))).astype(np.float32)
    if degenerate == "same_x":
        src[:, 0] = np.float32(scale / 3)
    elif degenerate == "same_point":
        dst[:] = dst[0]
    a, b = O.prepare(src, dst), native.host_prepare(src, dst)
    for k in ("N1", "N2", "C1", "C2", "iC2", "iN2", "nf1", "nf2", "cf1", "cf2"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k


@settings(max_examples=15, deadline=None)
@given(n=st.integers(5, 200), rows=st.integers(1, 4), cols=st.integers(1, 4), seed=st.integers(0, 10_000),
       gamma=st.sampled_from([0.1, 0.3, 0.8]), sigma=st.sampled_from([5.0, 30.0, 300.0]))
def test_oracle_loop_and_fast_agree(n, rows, cols, seed, gamma, sigma):
    """SVD of the weighted system (the reference's route) and the eigen-decomposition of its
    normal matrix (the route of the vectorised oracle and of the GPU engine) agree whenever the
    clamp gamma keeps the system determined.  (With gamma = 0 and sigma far below the keypoint
    spacing a cell can see a single keypoint: singular values 3.6, 1.1, 8e-4 ... 9e-8 - no method
    defines a homography there, and the two routes differ by 1e-4 px and more.)"""
    rng = np.random.default_rng(seed)
    src = (rng.random((n, 2)) * [320, 240]).astype(np.float32)
    dst = (src @ np.array([[1.02, -0.03], [0.02, 0.97]], np.float32) + np.float32([5, -3])
           + rng.normal(0, 0.5, (n, 2)).astype(np.float32)).astype(np.float32)
    verts = rng.random((rows, cols, 2)) * [320, 240]
    Hl, Wl = O.local_homography_loop(src, dst, verts, gamma, sigma)
    Hf, Wf = O.local_homography_fast(src, dst, verts, gamma, sigma, want_weights=True)
    assert np.array_equal(Wl, Wf)
    assert O.reprojection_rmse_delta(Hl, Hf, src).max() < 1e-5
