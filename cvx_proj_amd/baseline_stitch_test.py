"""The correspondence / seed-homography step that the reference's ``apap.py __main__`` imports
from ``pyviz/baseline_stitch_test.py`` (apap.py:17,238), without OpenCV.

``visualize_feature_pairs`` keeps the reference's signature and return convention
(baseline_stitch_test.py:18-61).  What differs, because the OpenCV pieces are not available:

* the reference re-describes the keypoints of ``keypoints.mat`` with SIFT and re-matches them
  with FLANN (utils.py:142-151).  Here column k of the centre picture's keypoints is taken as
  the match of column k of the other picture's - the file stores them as correspondences
  (utils.py:63-66 reads both from the same 6 x n matrix);
* ``cv.findHomography(..., cv.RANSAC, 5.0)`` is replaced by this repository's GPU RANSAC
  (``apap_find_homography_ransac``: same contract, own sampler, re-fit by the hot path's
  normalised DLT; see ``oracle/frontend_oracle.py``);
* nothing is drawn or shown (``disp``, ``drawMatches``, ``imwrite`` of the match picture).
"""
from __future__ import annotations

import numpy as np

from . import _native
from .utils import DEFAULT_ROOT, get_features, save2mat

CENTER_PIC_ID = 3           # baseline_stitch_test.py:14

__all__ = ["visualize_feature_pairs", "find_homography", "CENTER_PIC_ID"]


def find_homography(src_pts, dst_pts, thresh=5.0, device=-1, **kw):
    """``cv.findHomography(src_pts, dst_pts, cv.RANSAC, thresh)`` -> ``(H or None, mask (n, 1) uint8)``."""
    return _native.find_homography_ransac(src_pts, dst_pts, thresh, device=device, **kw)


def visualize_feature_pairs(center_img, other_img, case_idx: int = 1, pic_id: int = 1, disp=False, swap=True,
                            savemat=False, root: str = DEFAULT_ROOT, device=-1):
    """baseline_stitch_test.py:18-61.  Returns ``(dst_pts, src_pts, inv(H))`` when ``swap`` (the
    form apap.py:238 uses: points of the other picture first, H maps other -> centre), else
    ``(src_pts, dst_pts, H)``; the points are the RANSAC inliers, float32."""
    raw_kpts_cp, raw_kpts_op = get_features(case_idx, pic_id, CENTER_PIC_ID, root=root)
    src_pts = np.float32(raw_kpts_cp)          # centre picture ("query")
    dst_pts = np.float32(raw_kpts_op)          # other picture ("train")
    print(f"Coarse matching result: {len(src_pts)}")
    H, mask = find_homography(src_pts, dst_pts, 5.0, device=device)
    if H is None:
        raise np.linalg.LinAlgError("no homography: fewer than 4 consistent correspondences")
    if swap:
        H = np.linalg.inv(H)
    print(f"Number of matches: {len(src_pts)}, valid matches: {mask.sum()}")
    keep = mask.ravel() > 0
    src_pts, dst_pts = src_pts[keep], dst_pts[keep]
    if savemat:
        save2mat("matched_cp", src_pts)
        save2mat("matched_op", dst_pts)
    if swap:
        return dst_pts, src_pts, H
    return src_pts, dst_pts, H
