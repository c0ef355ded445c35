"""Host-side logic (no GPU): the C set-up code against the oracle bit for bit, the
Python mirror's helpers against the reference's golden vectors, geometry, synthetic
inputs, config reader."""
import numpy as np
import pytest

from oracle import apap_oracle as O
from cvx_proj_amd import geometry, synth
from cvx_proj_amd.apap import APAP, read_config

MOMENT_INDEX = ([(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
                + [(i, 6 + j) for i in range(3) for j in range(3)]
                + [(3 + i, 6 + j) for i in range(3) for j in range(3)]
                + [(6, 6), (6, 7), (6, 8), (7, 7), (7, 8), (8, 8)])


@pytest.mark.parametrize("name", ["tiny_sigma100", "tiny_sigma6"])
def test_c_prepare_matches_reference_vectors(native, golden, name):
    g = golden(name)
    q = native.host_prepare(g["src"], g["dst"])
    for k in ("N1", "N2", "C1", "C2", "nf1", "nf2", "cf1", "cf2"):
        assert np.array_equal(q[k], g[k]), k
    assert np.array_equal(native.host_dlt_rows(q["cf1"], q["cf2"]), g["aa"])
    assert np.array_equal(q["iC2"], np.linalg.inv(g["C2"]))
    assert np.array_equal(q["iN2"], np.linalg.inv(g["N2"]))


def test_prepare_beyond_numpy_reduction_buffer_vs_reference(native, golden):
    """n = 8192, 8193, 20001, 50000 through the reference's own set-up functions
    (tests/golden/make_golden.py prepare): the oracle and the C host code give the same four
    matrices and the same normalised points and DLT rows, bit for bit."""
    import hashlib
    g = golden("prepare_ref")
    sha = lambda a: np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)  # noqa: E731
    for n in (int(v) for v in g["sizes"]):
        rng = np.random.default_rng(900 + n)
        src = (rng.random((n, 2)) * [3840, 2160]).astype(np.float32)
        dst = (src + rng.normal(0, 5, (n, 2)) + [30, -20]).astype(np.float32)
        p = O.prepare(src, dst)
        q = native.host_prepare(src, dst)
        for who, r in (("oracle", p), ("C host", q)):
            for k in ("N1", "N2", "C1", "C2"):
                assert np.array_equal(r[k], g[f"{k}_{n}"]), (who, n, k)
            for k in ("nf1", "nf2", "cf1", "cf2"):
                assert np.array_equal(sha(r[k]), g[f"{k}_{n}"]), (who, n, k)
        assert np.array_equal(sha(p["aa"]), g[f"aa_{n}"])
        assert np.array_equal(sha(native.host_dlt_rows(q["cf1"], q["cf2"])), g[f"aa_{n}"])


def test_c_prepare_matches_oracle_random(native):
    rng = np.random.default_rng(3)
    # beyond 8192 numpy's reductions run in buffer-sized pieces: cover that regime too
    sizes = ([2, 3, 7, 8, 9, 127, 128, 129, 1000, 2000, 5000, 8191, 8192, 8193, 10000, 16385, 20001, 40000]
             + [int(v) for v in rng.integers(10, 6000, 30)] + [int(v) for v in rng.integers(8193, 50000, 10)])
    for t, n in enumerate(sizes):
        w, h = rng.choice([200, 1920, 3840, 7680]), rng.choice([140, 1080, 2160, 4320])
        src = (rng.random((n, 2)) * [w, h]).astype(np.float32)
        dst = (src + rng.normal(0, 5, (n, 2)) + [30, -20]).astype(np.float32)
        if t == 5:
            src[:, 0] = 5.0            # zero spread along x: the std == 0 guard of apap.py:81-82
        p = O.prepare(src, dst)
        q = native.host_prepare(src, dst)
        for k in ("N1", "N2", "C1", "C2", "iC2", "iN2", "nf1", "nf2", "cf1", "cf2"):
            assert np.array_equal(p[k], q[k]), (n, k)
        aa = native.host_dlt_rows(q["cf1"], q["cf2"])
        assert np.array_equal(aa, p["aa"])
        table = native.host_build_table(src, q["cf1"], q["cf2"])
        P = O.moments_from_rows(aa)
        assert np.array_equal(table[:, :30], np.stack([P[:, i, j] for i, j in MOMENT_INDEX], axis=1))
        assert np.array_equal(table[:, 30:], src.astype(np.float64))
        # the block structure the kernels rely on
        assert np.array_equal(P[:, 3:6, 3:6], P[:, 0:3, 0:3]) and not P[:, 0:3, 3:6].any()
        # the opt-in 24-sum table (APAP_OPT_MOMENTS = 24): exact float64 products of the rows' float32 entries
        t24 = native.host_build_table(src, q["cf1"], q["cf2"], moments=24)
        a64 = aa.astype(np.float64)
        x, y, c, f = a64[0::2, 0], a64[0::2, 1], a64[0::2, 8], a64[1::2, 8]
        pp = np.stack([x * x, x * y, x, y * y, y, np.ones_like(x)], axis=1)
        r = c * c + f * f
        assert np.array_equal(t24[:, :24], np.concatenate([pp, c[:, None] * pp, f[:, None] * pp, r[:, None] * pp], axis=1))
        assert np.array_equal(t24[:, 24:28], np.stack([a64[0::2, 6], a64[0::2, 7], a64[1::2, 6], a64[1::2, 7]], axis=1))
        assert (t24[:, 28].view(np.int64) == 0x7ff8242424242424).all()
        assert np.array_equal(t24[:, 29].copy().view(np.float32).reshape(-1, 2), src) and np.array_equal(t24[:, 30:], src.astype(np.float64))
        # ... of which the normal matrix is the block form of SURVEY.md section 8a, equal to the 30-sum one up to the float32
        # rounding of the reference's products
        S = t24[:, :24].sum(axis=0)
        M = np.zeros((9, 9))
        tri = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
        for u, (i, j) in enumerate(tri):
            for (bi, bj, k) in ((0, 0, 0), (1, 1, 0), (0, 2, 1), (1, 2, 2), (2, 2, 3)):
                M[3 * bi + i, 3 * bj + j] = M[3 * bi + j, 3 * bj + i] = S[6 * k + u]
        M = np.triu(M) + np.triu(M, 1).T
        assert np.allclose(M, P.sum(axis=0), rtol=0, atol=3e-7 * np.abs(P.sum(axis=0)).max())
        den = native.host_build_denorm(q["iC2"], q["C1"], q["iN2"], q["N1"])
        assert np.array_equal(den, np.concatenate([q[k].astype(np.float64).ravel() for k in ("iC2", "C1", "iN2", "N1")]))


def test_singular_normaliser_is_reported(native):
    # all keypoints identical: mean distance 0 -> scale = sqrt(2)/1e-8, still invertible;
    # a NaN keypoint makes the pivots NaN, not zero - numpy raises nothing either.
    src = np.full((4, 2), 3.0, np.float32)
    q = native.host_prepare(src, src)
    assert np.isfinite(q["N1"]).all() and q["C1"][0, 0] == np.float32(np.sqrt(2))


@pytest.mark.parametrize("name", ["tiny_sigma100", "tiny_sigma6"])
def test_python_mirror_helpers(native, golden, name):
    g = golden(name)
    N1, nf1 = APAP.getNormalize2DPts(g["src"])
    N2, nf2 = APAP.getNormalize2DPts(g["dst"])
    assert np.array_equal(N1, g["N1"]) and np.array_equal(nf1, g["nf1"])
    assert np.array_equal(N2, g["N2"]) and np.array_equal(nf2, g["nf2"])
    C1 = APAP.getConditionerFromPts(nf1)
    C2 = APAP.getConditionerFromPts(nf2)
    assert np.array_equal(C1, g["C1"]) and np.array_equal(C2, g["C2"])
    cf1 = APAP.point_normalize(nf1, C1)
    cf2 = APAP.point_normalize(nf2, C2)
    assert np.array_equal(cf1, g["cf1"]) and np.array_equal(cf2, g["cf2"])
    assert np.array_equal(APAP.matrix_generate(len(cf1), cf1, cf2), g["aa"])
    t = APAP.warp_coordinate_estimate(np.array([3, 4, 1]), g["Hinv_ref"][0, 0])
    assert t.dtype == np.float64 and t[2] == 1.0


@pytest.mark.parametrize("name", ["tiny_sigma100", "tiny_sigma6"])
def test_geometry_matches_reference(golden, name):
    g = golden(name)
    fw, fh, ox, oy = (int(v) for v in g["final"])

    class S:
        shape = g["img"].shape

    assert tuple(int(v) for v in geometry.final_size(S, S, g["Hg"])) == (fw, fh, ox, oy)
    assert np.array_equal(geometry.get_mesh((fw, fh), 6), g["mesh"])
    assert np.array_equal(geometry.get_vertice((fw, fh), 5, (ox, oy)), g["vertices"])
    assert np.array_equal(geometry.uniform_blend(g["warped_ref"], g["blend_other"]), g["blended_ref"])


def test_synth_is_deterministic_and_matches_golden_canvas(golden):
    a = synth.config_pair("C2", with_image=False)
    b = synth.config_pair("C2", with_image=False)
    assert np.array_equal(a.src, b.src) and np.array_equal(a.dst, b.dst)
    assert (a.final_w, a.final_h, a.off_x, a.off_y) == tuple(int(v) for v in golden("c2_ref")["final"])
    assert a.vertices.shape == (100, 100, 2) and a.mesh.shape == (2, 101)
    c = synth.config_pair("C1", with_image=True)
    assert c.img.shape == (768, 768, 3) and c.img.dtype == np.uint8


def test_config_reader(tmp_path):
    f = tmp_path / "case1.txt"
    f.write_text("em_steps = 2\naffinity_eps = 30.0\nmesh_size = 40\nsigma = 12.5  # comment\n")
    cfg = read_config(str(f))
    assert cfg["mesh_size"] == "40" and float(cfg["sigma"]) == 12.5 and cfg["em_steps"] == "2"


@pytest.mark.parametrize("tag", ["tiny", "tiny6", "mixa", "mixb", "ints", "c2", "big"])
def test_c_prepare_in_the_dtype_of_the_keypoints(native, golden, tag):
    """float64 keypoints are NOT narrowed - the C set-up runs in the dtype of each point set, as the
    reference's functions do (apap.py:35-100), and reproduces the reference's matrices, normalised points and DLT rows bit
    for bit (float64 sets, one float64 set beside a float32 one, int64 keypoints, 20 001 keypoints)."""
    import hashlib
    sha = lambda a: np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)  # noqa: E731
    g = golden("f64pts_ref")
    src, dst = g[f"{tag}_src"], g[f"{tag}_dst"]
    q = native.host_prepare(src, dst)
    aa = native.host_dlt_rows(q["cf1"], q["cf2"])
    for k in ("N1", "N2", "C1", "C2"):
        assert np.array_equal(q[k], g[f"{tag}_{k}"]), k
    assert np.array_equal(q["iC2"], np.linalg.inv(g[f"{tag}_C2"])) and np.array_equal(q["iN2"], np.linalg.inv(g[f"{tag}_N2"]))
    for k, v in (("nf1", q["nf1"]), ("nf2", q["nf2"]), ("cf1", q["cf1"]), ("cf2", q["cf2"]), ("aa", aa)):
        if f"{tag}_{k}" in g:
            assert v.dtype == g[f"{tag}_{k}"].dtype and np.array_equal(v, g[f"{tag}_{k}"]), k
        else:
            assert np.array_equal(sha(v), g[f"{tag}_{k}_sha"]), k
    # the device table: moments of the DLT rows + the source keypoints as float64, untouched
    table = native.host_build_table(src, q["cf1"], q["cf2"])
    assert np.array_equal(table[:, 30:32], np.asarray(src, np.float64))
    r = aa.astype(np.float64).reshape(-1, 2, 9)
    m = np.einsum("kri,krj->kij", r, r)
    want = np.stack([m[:, i, j] for i, j in MOMENT_INDEX], axis=1)
    assert np.array_equal(table[:, :30], want)


def test_mat_writer_equals_scipy(tmp_path):
    """cvx_proj_amd.utils.savemat_f64 - the `.mat` file of apap.py:264 without importing scipy - writes what scipy.io.savemat does,
    byte for byte behind the creation time of the 128-byte header, and loads back through scipy."""
    import re
    import scipy.io
    from cvx_proj_amd import utils as U
    rng = np.random.default_rng(3)
    for shape, name in (((400, 9), "H"), ((10000, 9), "H"), ((1, 1), "H"), ((7, 3), "sift_feature"), ((5, 2), "abcd"), ((5, 2), "abcde")):
        a = rng.normal(size=shape)
        for arr in (a, np.asfortranarray(a), a[::-1]):
            U.save2mat("mine", arr, name=name, prefix=str(tmp_path) + "/")
            scipy.io.savemat(str(tmp_path / "theirs.mat"), {name: arr})
            mine, theirs = (tmp_path / "mine.mat").read_bytes(), (tmp_path / "theirs.mat").read_bytes()
            assert mine[128:] == theirs[128:] and mine[116:128] == theirs[116:128]
            strip = lambda b: re.sub(rb"Created on: .*", b"", b[:116].rstrip(b"\0"))      # noqa: E731
            assert strip(mine) == strip(theirs)
            assert np.array_equal(scipy.io.loadmat(str(tmp_path / "mine.mat"))[name], arr)
    # anything that is not a real float64 matrix still goes to scipy
    U.save2mat("ints", np.arange(6, dtype=np.int32).reshape(2, 3), name="k", prefix=str(tmp_path) + "/")
    assert scipy.io.loadmat(str(tmp_path / "ints.mat"))["k"].dtype == np.int32
