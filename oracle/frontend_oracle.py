"""CPU oracle for the callers on either side of the APAP hot path (TEST INFRASTRUCTURE, not
product): the image pre-processing and the correspondence / seed-homography front end that
the reference's ``apap.py __main__`` runs before ``APAP.local_homography`` (SURVEY.md 8f,
ranks 4 and 1).  Same rules of use as ``apap_oracle.py``: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

PARITY UNPINNED against OpenCV for everything in this file.  The reference obtains these
steps from ``opencv-python`` 4.6.0.66 (``README.md:16``, ``requirements.txt:3``), which is
absent from ``/root/reference`` and from this image, and the reference holds no test or golden
vector for them:

* ``equalize_hist_channel`` restates the published algorithm of ``cv::equalizeHist``
  (OpenCV 4.6.0 ``modules/imgproc/src/histogram.cpp``): 256-bin histogram; ``i0`` = first
  non-empty bin; a constant image is returned as is; otherwise
  ``scale = 255.f / (total - hist[i0])`` (float32), ``lut[i0] = 0`` and for ``i > i0``
  ``lut[i] = saturate_cast<uchar>(sum(hist[i0+1 .. i]) * scale)`` with the running sum an
  ``int`` converted to float32 and ``saturate_cast`` rounding half to even (``cvRound``).
  Call site: ``utils.py:85-91`` (per channel, ``np.stack(..., axis=-1)``).
* ``ransac_homography`` is NOT a restatement of ``cv::findHomography(..., RANSAC, 5.0)``
  (``baseline_stitch_test.py:42``): OpenCV's sampler state, its adaptive iteration count and
  its Levenberg-Marquardt polish are not reproduced.  It is the specification of this
  repository's own estimator with the same contract - 4-point hypotheses, forward
  reprojection error against the 5-pixel threshold, the first hypothesis with the most
  inliers wins, the model is then re-fitted to its inliers - and pins the HIP implementation
  bit for bit (sampler, minimal solver, inlier counts, mask).
"""
from __future__ import annotations

import numpy as np

__all__ = ["equalize_hist_channel", "equalize_hist_image", "equalize_lut"]


def equalize_lut(hist, total):
    """256-entry lookup table of ``cv::equalizeHist`` from a histogram; ``None`` for a
    constant image (OpenCV then fills the output with that value)."""
    hist = np.asarray(hist, dtype=np.int64)
    i0 = int(np.flatnonzero(hist)[0])
    if hist[i0] == total:
        return None
    scale = np.float32(255.0) / np.float32(int(total) - int(hist[i0]))
    lut = np.zeros(256, dtype=np.uint8)
    sums = np.cumsum(hist[i0 + 1:], dtype=np.int64)
    # int -> float32 (round to nearest even), float32 product, cvRound (half to even), clamp
    v = np.rint(sums.astype(np.int32).astype(np.float32) * scale)
    lut[i0 + 1:] = np.clip(v, 0, 255).astype(np.uint8)
    return lut


def equalize_hist_channel(channel):
    """``cv.equalizeHist`` of one uint8 plane (any shape)."""
    channel = np.ascontiguousarray(channel, dtype=np.uint8)
    hist = np.bincount(channel.ravel(), minlength=256)
    lut = equalize_lut(hist, channel.size)
    if lut is None:
        return channel.copy()
    return lut[channel]


def equalize_hist_image(img):
    """``np.stack([cv.equalizeHist(img[..., i]) for i in range(3)], axis=-1)``
    (reference utils.py:88)."""
    return np.stack([equalize_hist_channel(img[..., i]) for i in range(img.shape[-1])], axis=-1)


# --------------------------------------------------------------------------------------
# Seed homography: this repository's RANSAC estimator (see the module header: a
# specification pinned bit for bit to the HIP kernels, not a restatement of OpenCV's).
# Contract of the call it replaces: baseline_stitch_test.py:42
#     H, mask = cv.findHomography(src_pts, dst_pts, cv.RANSAC, 5.0)
# --------------------------------------------------------------------------------------
RANSAC_ITERATIONS = 2048
RANSAC_SEED = 0x5EEDC0DE5EEDC0DE
_M64 = (1 << 64) - 1


def splitmix64(x):
    """One output of the splitmix64 sequence for state ``x`` (uint64 array or int)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def ransac_sample(n, iterations=RANSAC_ITERATIONS, seed=RANSAC_SEED):
    """(iterations, 4) distinct indices per hypothesis.  Draw j of hypothesis h uses the high 32
    bits of ``splitmix64(seed + 4 h + j)`` modulo ``n - j``, then steps over the earlier picks
    in ascending order (sampling without replacement, integer arithmetic only)."""
    assert n >= 4
    h = np.arange(iterations, dtype=np.uint64)
    picks = np.zeros((iterations, 4), dtype=np.int64)
    with np.errstate(over="ignore"):
        for j in range(4):
            r = (splitmix64(np.uint64(seed) + h * np.uint64(4) + np.uint64(j)) >> np.uint64(32)).astype(np.int64)
            p = r % (n - j)
            prev = np.sort(picks[:, :j], axis=1)
            for k in range(j):
                p = p + (p >= prev[:, k])
            picks[:, j] = p
    return picks


def ransac_minimal_solve(src4, dst4):
    """Homographies (h22 = 1) through 4 correspondences each: ``src4``, ``dst4`` (K, 4, 2)
    float64.  8 x 8 Gaussian elimination with partial pivoting (first largest pivot), every
    product and difference rounded separately, back substitution in ascending column order.
    Returns (K, 9); rows of singular systems are NaN."""
    K = src4.shape[0]
    x, y = src4[..., 0], src4[..., 1]
    u, v = dst4[..., 0], dst4[..., 1]
    a = np.zeros((K, 8, 9))
    one, zero = np.ones_like(x), np.zeros_like(x)
    a[:, 0::2] = np.stack([x, y, one, zero, zero, zero, -(u * x), -(u * y), u], axis=-1)
    a[:, 1::2] = np.stack([zero, zero, zero, x, y, one, -(v * x), -(v * y), v], axis=-1)
    ok = np.ones(K, dtype=bool)
    rows = np.arange(K)
    with np.errstate(all="ignore"):
        for c in range(8):
            p = c + np.argmax(np.abs(a[:, c:, c]), axis=1)          # first largest
            tmp = a[rows, p].copy()
            a[rows, p] = a[:, c]
            a[:, c] = tmp
            piv = a[:, c, c]
            ok &= (piv != 0) & np.isfinite(piv)
            for r in range(c + 1, 8):
                f = a[:, r, c] / piv
                a[:, r, c:] = a[:, r, c:] - f[:, None] * a[:, c, c:]
        h = np.zeros((K, 8))
        for i in range(7, -1, -1):
            s = a[:, i, 8].copy()
            for j in range(i + 1, 8):
                s = s - a[:, i, j] * h[:, j]
            h[:, i] = s / a[:, i, i]
    out = np.concatenate([h, np.ones((K, 1))], axis=1)
    out[~ok] = np.nan
    return out


def ransac_errors(H9, src, dst):
    """Squared forward reprojection error of every point under every hypothesis, (K, n)."""
    x, y = src[:, 0].astype(np.float64), src[:, 1].astype(np.float64)
    u, v = dst[:, 0].astype(np.float64), dst[:, 1].astype(np.float64)
    h = H9[:, :, None]
    with np.errstate(all="ignore"):
        w = (h[:, 6] * x + h[:, 7] * y) + h[:, 8]
        px = ((h[:, 0] * x + h[:, 1] * y) + h[:, 2]) / w
        py = ((h[:, 3] * x + h[:, 4] * y) + h[:, 5]) / w
        dx, dy = px - u, py - v
        return dx * dx + dy * dy


def ransac_core(src, dst, thresh=5.0, iterations=RANSAC_ITERATIONS, seed=RANSAC_SEED):
    """Hypotheses, inlier counts, and the first hypothesis with the most inliers.
    Returns ``dict(picks, H, counts, best, count, mask)``; ``mask`` is that hypothesis's
    inlier mask (uint8, n)."""
    src = np.ascontiguousarray(src, dtype=np.float32)
    dst = np.ascontiguousarray(dst, dtype=np.float32)
    picks = ransac_sample(len(src), iterations, seed)
    H = ransac_minimal_solve(src[picks].astype(np.float64), dst[picks].astype(np.float64))
    with np.errstate(invalid="ignore"):
        inl = ransac_errors(H, src, dst) <= thresh * thresh          # NaN compares false
    counts = inl.sum(axis=1).astype(np.int64)
    best = int(np.argmax(counts))                                    # first maximum
    return dict(picks=picks, H=H, counts=counts, best=best, count=int(counts[best]), mask=inl[best].astype(np.uint8))


def ransac_homography(src, dst, thresh=5.0, iterations=RANSAC_ITERATIONS, seed=RANSAC_SEED):
    """``(H, mask)`` with the contract of ``cv.findHomography(src, dst, cv.RANSAC, thresh)``:
    H float64 3 x 3 with H[2, 2] = 1 re-fitted to the inliers by the normalised DLT of the hot
    path itself (all weights 1: apap.py:35-119,160-168 on one cell), mask (n, 1) uint8;
    ``(None, zeros)`` when fewer than 4 points agree."""
    from . import apap_oracle as O
    core = ransac_core(src, dst, thresh, iterations, seed)
    mask = core["mask"].reshape(-1, 1)
    if core["count"] < 4:
        return None, np.zeros_like(mask)
    keep = core["mask"].astype(bool)
    s = np.ascontiguousarray(src, dtype=np.float32)[keep]
    d = np.ascontiguousarray(dst, dtype=np.float32)[keep]
    H, _ = O.local_homography_loop(s, d, np.zeros((1, 1, 2)), 1.0, 1.0, want_weights=False)
    return H[0, 0].astype(np.float64), mask
