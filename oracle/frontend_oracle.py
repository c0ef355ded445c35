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
