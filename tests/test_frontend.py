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
        native.check(native.lib().apap_equalize_hist_device(buf_in.data_ptr() + off_in, h, w, c,
                                                            buf_out.data_ptr() + off_out, work.data_ptr(),
                                                            work.numel(), ctypes.c_void_p(0)))
        torch.cuda.synchronize()
        got = buf_out[off_out:off_out + nbytes].cpu().numpy().reshape(h, w, c)
        assert np.array_equal(got, ref), (off_in, off_out)
        assert int(buf_out[:off_out].sum()) == 0 and int(buf_out[off_out + nbytes:].sum()) == 0   # nothing outside
    rc = native.lib().apap_equalize_hist_device(buf_in.data_ptr(), h, w, 5, buf_out.data_ptr(), work.data_ptr(),
                                                work.numel(), ctypes.c_void_p(0))
    assert rc == native.ERR_INVALID_ARG
    rc = native.lib().apap_equalize_hist_device(buf_in.data_ptr(), h, w, c, buf_out.data_ptr(), work.data_ptr(),
                                                8, ctypes.c_void_p(0))
    assert rc == native.ERR_WORKSPACE
