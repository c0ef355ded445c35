"""The callers of the hot path: histogram equalisation and the keypoint loader.
CPU tests pin the oracle's restatement of ``cv::equalizeHist`` to an independent scalar
transcription of the published algorithm and check the host-side loaders; GPU tests compare
the HIP kernels with the oracle byte for byte through the C ABI."""
import ctypes
import os

import numpy as np
import pytest

from oracle import frontend_oracle as F


def scalar_equalize(plane):
    """cv::equalizeHist transcribed statement by statement (histogram.cpp), float32 scalars."""
    flat = plane.ravel()
    hist = [0] * 256
    for v in flat.tolist():
        hist[v] += 1
    i = 0
    while not hist[i]:
        i += 1
    total = flat.size
    if hist[i] == total:
        return np.full_like(plane, i)
    scale = np.float32(255.0) / np.float32(total - hist[i])
    lut = [0] * 256
    s = 0
    i += 1
    while i < 256:
        s += hist[i]
        r = int(np.rint(np.float32(s) * scale))            # saturate_cast<uchar>(float) = cvRound + clamp
        lut[i] = min(max(r, 0), 255)
        i += 1
    return np.array(lut, dtype=np.uint8)[plane]


def natural_like(h, w, c, seed):
    """Low-contrast gradient + noise: heavy histogram bins, unlike uniform noise."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = 90 + 40 * np.sin(x / 37.0) * np.cos(y / 23.0)
    img = base[..., None] + rng.normal(0, 6, (h, w, c)) + np.arange(c) * 9
    return np.clip(img, 0, 255).astype(np.uint8)


# ------------------------------------------------------------------ CPU: the oracle itself
@pytest.mark.parametrize("seed", range(4))
def test_oracle_equalize_matches_scalar_transcription(seed):
    rng = np.random.default_rng(seed)
    for plane in (rng.integers(0, 256, (37, 53), dtype=np.uint8),
                  rng.integers(100, 110, (20, 31), dtype=np.uint8),
                  natural_like(64, 80, 1, seed)[..., 0]):
        assert np.array_equal(F.equalize_hist_channel(plane), scalar_equalize(plane))


def test_oracle_equalize_known_answers():
    # two grey levels, 3 : 1 -> the lower maps to 0, the upper to 255
    plane = np.array([[10, 10, 10, 200]], dtype=np.uint8)
    assert F.equalize_hist_channel(plane).tolist() == [[0, 0, 0, 255]]
    # four equally frequent levels -> 0, 85, 170, 255
    plane = np.array([[1, 2, 3, 4] * 5], dtype=np.uint8)
    assert F.equalize_hist_channel(plane).tolist() == [[0, 85, 170, 255] * 5]
    # a constant plane is returned unchanged
    plane = np.full((5, 7), 42, dtype=np.uint8)
    assert np.array_equal(F.equalize_hist_channel(plane), plane)
    # ties of cvRound go to even: scale = 255/6, the running sum 1 gives 42.5 -> 42 and 3 gives 127.5 -> 128
    plane = np.array([[0, 1, 2, 2, 3, 4, 4]], dtype=np.uint8)
    assert F.equalize_hist_channel(plane).tolist() == [[0, 42, 128, 128, 170, 255, 255]]


def test_oracle_equalize_image_is_per_channel():
    img = natural_like(40, 56, 3, 7)
    out = F.equalize_hist_image(img)
    for c in range(3):
        assert np.array_equal(out[..., c], F.equalize_hist_channel(img[..., c]))
    # equalisation is idempotent on its own histogram support ordering: monotone in the input
    for c in range(3):
        order = np.argsort(img[..., c].ravel(), kind="stable")
        assert (np.diff(out[..., c].ravel()[order].astype(int)) >= 0).all()


# ------------------------------------------------------------------ CPU: host-side loaders
def write_case(root, case_idx=1, n=40, seed=0, shape=(48, 64, 3)):
    """A synthetic dataset directory with the reference's layout (utils.py:27-66)."""
    import scipy.io
    from PIL import Image
    rng = np.random.default_rng(seed)
    base = os.path.join(root, f"case{case_idx}")
    for sub in ("scat", "no_scat"):
        os.makedirs(os.path.join(base, sub), exist_ok=True)
    imgs = {}
    for i in range(1, 6):
        for sub, stem in (("scat", "img_haze"), ("no_scat", "img_nohaze")):
            rgb = rng.integers(0, 256, shape, dtype=np.uint8)
            Image.fromarray(rgb).save(os.path.join(base, sub, f"{stem}{i}.png"))
            imgs[(sub, i)] = rgb
    cells = np.empty((4, 1), dtype=object)
    mats = []
    for k in range(4):
        m = np.ones((6, n + k))
        m[0:2] = rng.random((2, n + k)) * [[shape[1]], [shape[0]]]
        m[3:5] = rng.random((2, n + k)) * [[shape[1]], [shape[0]]]
        cells[k, 0] = m
        mats.append(m)
    scipy.io.savemat(os.path.join(base, "keypoints.mat"), {"keypoints": cells})
    return imgs, mats


def test_get_features_indexing(tmp_path):
    from cvx_proj_amd import utils as U
    _, mats = write_case(str(tmp_path))
    for pic_id, k in ((1, 0), (2, 1), (4, 2), (5, 3)):       # picture 3 is the centre: utils.py:64
        cp, op = U.get_features(1, pic_id, 3, root=str(tmp_path))
        assert np.array_equal(cp, mats[k][:2].T) and np.array_equal(op, mats[k][3:5].T)
    with pytest.raises(ValueError):
        U.get_features(1, 3, 3, root=str(tmp_path))


def test_get_features_vs_reference(tmp_path, golden):
    """The reference's own reader (utils.py:55-66, run by tests/golden/make_golden.py) on the same
    keypoints.mat: same arrays for every valid picture id, same error for the centre id."""
    import scipy.io
    from cvx_proj_amd import utils as U
    g = golden("keypoints_ref")
    cells = np.empty((4, 1), dtype=object)
    for k in range(4):
        cells[k, 0] = g[f"mat{k}"]
    os.makedirs(tmp_path / "case1")
    scipy.io.savemat(tmp_path / "case1" / "keypoints.mat", {"keypoints": cells})
    for pic_id in (1, 2, 4, 5):
        cp, op = U.get_features(1, pic_id, 3, root=str(tmp_path))
        assert cp.dtype == g[f"cp{pic_id}"].dtype and np.array_equal(cp, g[f"cp{pic_id}"])
        assert np.array_equal(op, g[f"op{pic_id}"])


def test_imread_is_bgr_and_paths(tmp_path):
    from cvx_proj_amd import utils as U
    imgs, _ = write_case(str(tmp_path))
    assert U.get_path(2, 4) == "../diff_1/raw_data/case2/scat/img_haze4.png"
    assert U.get_path(2, 4, True) == "../diff_1/raw_data/case2/no_scat/img_nohaze4.png"
    got = U.imread(U.get_path(1, 2, root=str(tmp_path)))
    assert np.array_equal(got, imgs[("scat", 2)][..., ::-1])
    c, o = U.get_no_scat_img(1, 5, 3, root=str(tmp_path))
    assert np.array_equal(c, imgs[("no_scat", 3)][..., ::-1]) and np.array_equal(o, imgs[("no_scat", 5)][..., ::-1])
    assert U.imread(str(tmp_path / "missing.png")) is None


# ------------------------------------------------------------------ GPU: equalisation
@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 1, 3), (3, 5, 3), (37, 53, 3), (64, 64, 1), (48, 70, 4), (33, 19, 2), (480, 640, 3)])
def test_equalize_vs_oracle(native, shape):
    rng = np.random.default_rng(sum(shape))
    for img in (rng.integers(0, 256, shape, dtype=np.uint8), natural_like(*shape, seed=3)):
        out = native.equalize_hist(img)
        assert out.shape == img.shape and np.array_equal(out, F.equalize_hist_image(img))


@pytest.mark.gpu
def test_equalize_plane_constant_and_mixed(native):
    plane = natural_like(90, 120, 1, 5)[..., 0]
    assert np.array_equal(native.equalize_hist(plane), F.equalize_hist_channel(plane))
    img = natural_like(50, 66, 3, 6)
    img[..., 1] = 77                                   # one constant channel stays as it is
    out = native.equalize_hist(img)
    assert (out[..., 1] == 77).all() and np.array_equal(out, F.equalize_hist_image(img))
    img[..., 0] = np.where(img[..., 0] > 100, 255, 0)  # two levels only
    assert np.array_equal(native.equalize_hist(img), F.equalize_hist_image(img))


@pytest.mark.gpu
def test_equalize_full_4k_and_properties(native):
    from cvx_proj_amd.synth import config_pair
    img = config_pair("C3").img
    out = native.equalize_hist(img)
    assert np.array_equal(out, F.equalize_hist_image(img))
    again = native.equalize_hist(out)                  # an equalised uniform-noise image is a fixed point
    assert np.array_equal(again, F.equalize_hist_image(out))


@pytest.mark.gpu
def test_equalize_8k_running_sums_beyond_float32_integers(native):
    """33 Mpixel planes: the running sums pass 2^24, so the int -> float32 conversion of
    cv::equalizeHist rounds (to nearest even, on both sides)."""
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, (4320, 7680, 3), dtype=np.uint8)
    img[..., 2] = (img[..., 2] >> 3) + 40          # a narrow band: large bins, sums far from multiples of 2^k
    out = native.equalize_hist(img)
    assert np.array_equal(out, F.equalize_hist_image(img))


@pytest.mark.gpu
def test_equalize_device_entry_any_alignment(native):
    """Resident data, pointers at odd byte offsets (the kernels split head / 16-byte body / tail)."""
    import torch
    dev = torch.device("cuda:0")
    h, w, c = 61, 47, 3
    img = natural_like(h, w, c, 11)
    nbytes = img.size
    work = torch.zeros(native.lib().apap_equalize_workspace_bytes(c), dtype=torch.uint8, device=dev)
    ref = F.equalize_hist_image(img)
    for off_in, off_out in ((0, 0), (1, 0), (5, 3), (15, 9)):
        buf_in = torch.zeros(nbytes + 32, dtype=torch.uint8, device=dev)
        buf_out = torch.zeros(nbytes + 32, dtype=torch.uint8, device=dev)
        buf_in[off_in:off_in + nbytes] = torch.from_numpy(img.ravel()).to(dev)
        native.check(native.lib().apap_equalize_hist_device(None, buf_in.data_ptr() + off_in, h, w, c,
                                                            buf_out.data_ptr() + off_out, work.data_ptr(),
                                                            work.numel(), ctypes.c_void_p(0)))
        torch.cuda.synchronize()
        got = buf_out[off_out:off_out + nbytes].cpu().numpy().reshape(h, w, c)
        assert np.array_equal(got, ref), (off_in, off_out)
        assert int(buf_out[:off_out].sum()) == 0 and int(buf_out[off_out + nbytes:].sum()) == 0   # nothing outside
    rc = native.lib().apap_equalize_hist_device(None, buf_in.data_ptr(), h, w, 5, buf_out.data_ptr(), work.data_ptr(),
                                                work.numel(), ctypes.c_void_p(0))
    assert rc == native.ERR_INVALID_ARG
    rc = native.lib().apap_equalize_hist_device(None, buf_in.data_ptr(), h, w, c, buf_out.data_ptr(), work.data_ptr(),
                                                8, ctypes.c_void_p(0))
    assert rc == native.ERR_WORKSPACE


# ------------------------------------------------------------------ RANSAC seed homography
def ransac_case(n=600, outliers=0.3, noise=0.7, seed=0, size=(1920, 1080)):
    rng = np.random.default_rng(seed)
    Hg = np.array([[1.02, 0.01, 30.0], [-0.015, 0.99, -20.0], [1e-5, -2e-5, 1.0]])
    src = (rng.random((n, 2)) * size).astype(np.float32)
    q = np.c_[src.astype(np.float64), np.ones(n)] @ Hg.T
    dst = (q[:, :2] / q[:, 2:] + rng.normal(0, noise, (n, 2))).astype(np.float32)
    out = rng.random(n) < outliers
    dst[out] = (rng.random((int(out.sum()), 2)) * size).astype(np.float32)
    return src, dst, Hg, out


def degenerate_case(n=40):
    """Source points on one line (exactly, in float32): no 4 of them determine a homography."""
    t = np.arange(n, dtype=np.float32)
    line = np.stack([t, 2 * t], axis=1)
    other = (np.random.default_rng(5).random((n, 2)) * 1000).astype(np.float32)
    return line, other


def test_oracle_ransac_sampler_is_distinct_uniform_and_reproducible():
    for n in (4, 5, 9, 1000):
        p = F.ransac_sample(n, 4096)
        assert p.min() >= 0 and p.max() < n
        s = np.sort(p, axis=1)
        assert (s[:, 1:] != s[:, :-1]).all()
        assert np.array_equal(p, F.ransac_sample(n, 4096))
    counts = np.bincount(F.ransac_sample(16, 1 << 14).ravel(), minlength=16)
    assert counts.min() > 0.9 * counts.mean() and counts.max() < 1.1 * counts.mean()
    assert not np.array_equal(F.ransac_sample(50, 64, seed=1), F.ransac_sample(50, 64, seed=2))


def test_oracle_ransac_minimal_solver():
    rng = np.random.default_rng(1)
    Hg = np.array([[0.9, 0.05, 12.0], [-0.03, 1.1, -7.0], [2e-4, -1e-4, 1.0]])
    src4 = rng.random((64, 4, 2)) * 500
    q = np.concatenate([src4, np.ones((64, 4, 1))], axis=-1) @ Hg.T
    dst4 = q[..., :2] / q[..., 2:]
    H = F.ransac_minimal_solve(src4, dst4)
    assert np.allclose(H, Hg.ravel(), rtol=1e-6, atol=1e-7)
    # collinear source points: singular system -> NaN row, never an inlier
    src4[0] = [[0, 0], [1, 1], [2, 2], [3, 3]]
    H = F.ransac_minimal_solve(src4, dst4)
    assert np.isnan(H[0]).all() and np.isfinite(H[1:]).all()
    assert not (F.ransac_errors(H[:1], src4[1].astype(np.float32), dst4[1].astype(np.float32)) <= 25).any()


def test_oracle_ransac_recovers_inliers_and_model():
    src, dst, Hg, out = ransac_case()
    H, mask = F.ransac_homography(src, dst)
    assert mask.shape == (len(src), 1) and mask.dtype == np.uint8
    assert not (mask.ravel().astype(bool) & out).any()           # no gross outlier accepted
    assert mask.sum() >= 0.98 * (~out).sum()
    q = np.c_[src.astype(np.float64), np.ones(len(src))]
    a, b = q @ H.T, q @ Hg.T
    assert np.abs(a[:, :2] / a[:, 2:] - b[:, :2] / b[:, 2:]).max() < 0.5
    assert H[2, 2] == 1.0
    # collinear source points: every 4-point system is singular -> no model (cv returns None)
    line, other = degenerate_case()
    H, mask = F.ransac_homography(line, other, iterations=64)
    assert H is None and mask.sum() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("n,outliers,seed", [(600, 0.3, 0), (2000, 0.5, 1), (8, 0.0, 2), (4, 0.0, 3), (57, 0.2, 4)])
def test_ransac_device_half_is_bit_identical_to_the_oracle(native, n, outliers, seed):
    """Sampler, minimal solver, inlier counts, winner and mask: equal bit for bit."""
    import torch
    dev = torch.device("cuda:0")
    src, dst, _, _ = ransac_case(n, outliers, seed=seed)
    K = 512
    core = F.ransac_core(src, dst, 5.0, K, F.RANSAC_SEED)
    d_src, d_dst = torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev)
    wb = native.lib().apap_ransac_workspace_bytes(n, K)
    work = torch.zeros(wb, dtype=torch.uint8, device=dev)
    Hb = torch.zeros(9, dtype=torch.float64, device=dev)
    mask = torch.zeros(n, dtype=torch.uint8, device=dev)
    res = torch.zeros(2, dtype=torch.int32, device=dev)
    native.check(native.lib().apap_ransac_device(None, d_src.data_ptr(), d_dst.data_ptr(), n, 5.0, K, ctypes.c_ulonglong(F.RANSAC_SEED),
                                                 Hb.data_ptr(), mask.data_ptr(), res.data_ptr(), work.data_ptr(), wb,
                                                 ctypes.c_void_p(0)))
    torch.cuda.synchronize()
    H_all = work.cpu().numpy()[:K * 72].view(np.float64).reshape(K, 9)
    counts = work.cpu().numpy()[K * 72:K * 76].view(np.int32)
    assert np.array_equal(np.isnan(H_all), np.isnan(core["H"]))
    same = np.nan_to_num(H_all) == np.nan_to_num(core["H"])
    assert same.all(), f"{int((~same).sum())} of {same.size} hypothesis entries differ"
    assert np.array_equal(counts, core["counts"])
    assert res.cpu().tolist() == [core["best"], core["count"]]
    assert np.array_equal(mask.cpu().numpy(), core["mask"])
    assert np.array_equal(Hb.cpu().numpy(), core["H"][core["best"]])


@pytest.mark.gpu
def test_find_homography_vs_oracle_and_contract(native):
    src, dst, Hg, out = ransac_case(900, 0.35, seed=7)
    H, mask = native.find_homography_ransac(src, dst, 5.0)
    H_ref, mask_ref = F.ransac_homography(src, dst, 5.0)
    assert H.shape == (3, 3) and H.dtype == np.float64 and mask.shape == (900, 1) and mask.dtype == np.uint8
    assert np.array_equal(mask, mask_ref)
    assert np.array_equal(H, H_ref), np.abs(H - H_ref).max()      # the re-fit is the hot path: bit-exact float32
    assert not (mask.ravel().astype(bool) & out).any() and mask.sum() >= 0.98 * (~out).sum()
    # no model
    line, other = degenerate_case()
    H, mask = native.find_homography_ransac(line, other, iterations=64)
    assert H is None and mask.sum() == 0
    with pytest.raises(native.ApapError):
        native.find_homography_ransac(src[:3], dst[:3])


@pytest.mark.gpu
def test_cli_reference_flow_on_a_dataset_directory(native, tmp_path):
    """apap.py __main__ end to end (apap.py:236-265) on a synthetic dataset in the reference's
    layout: PNG read, equalisation, keypoints.mat, RANSAC seed, hot path, .mat output."""
    import scipy.io
    from PIL import Image
    from cvx_proj_amd import apap as A
    from oracle import apap_oracle as O
    root = tmp_path / "raw_data"
    base = root / "case1"
    (base / "scat").mkdir(parents=True)
    (base / "no_scat").mkdir()
    rng = np.random.default_rng(0)
    h, w = 240, 320
    for i in range(1, 6):
        for sub, stem in (("scat", "img_haze"), ("no_scat", "img_nohaze")):
            Image.fromarray(natural_like(h, w, 3, i)).save(base / sub / f"{stem}{i}.png")
    src, dst, Hg, out = ransac_case(300, 0.25, noise=0.4, seed=9, size=(w, h))
    cells = np.empty((4, 1), dtype=object)
    for k in range(4):
        m = np.ones((6, len(src)))
        m[0:2], m[3:5] = src.T, dst.T           # centre picture's points, other picture's points
        cells[k, 0] = m
    scipy.io.savemat(base / "keypoints.mat", {"keypoints": cells})
    out_prefix = str(tmp_path / "results") + "/"
    warp_file = str(tmp_path / "stitch.npy")
    assert A.main(["1", "2", "--data-root", str(root), "--mesh-size", "12", "--out-prefix", out_prefix,
                   "--stitch", warp_file]) == 0
    flat = scipy.io.loadmat(out_prefix + "case1/H32_apap.mat")["H"]
    assert flat.shape == (144, 9) and flat.dtype == np.float64
    # the same flow through the oracles
    H_ref, mask_ref = F.ransac_homography(src, dst, 5.0)
    keep = mask_ref.ravel() > 0
    Hswap = np.linalg.inv(H_ref)
    fsrc, fdst = dst[keep], src[keep]           # swap=True: other picture's points first
    pic = np.empty((h, w, 3), dtype=np.uint8)    # final_size only reads .shape (apap.py:240)
    fw, fh, ox, oy = O.final_size(pic, pic, Hswap)
    vertices = O.get_vertice((fw, fh), 12, (ox, oy))
    Hl, _ = O.local_homography_loop(fsrc, fdst, vertices, 0.5, 100, want_weights=False)
    assert np.array_equal(flat, O.invert_normalize_flatten(Hl))
    canvas = np.load(warp_file)
    assert canvas.shape == (fh, fw, 3) and canvas.any()
