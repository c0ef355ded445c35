"""The part of the reference's ``pyviz/utils.py`` that ``apap.py __main__`` calls before and
after the hot path (reference apap.py:17-18,236-238,245): image pre-processing, the
``keypoints.mat`` loader and the ``.mat`` writer - without OpenCV.

* ``visualize_equalized_hist`` / ``equalize_hist``: per-channel ``cv.equalizeHist`` on the GPU
  (``apap_equalize_hist``, two HIP kernels).  No CPU fallback.
* ``get_features``: same indexing of ``keypoints.mat`` as utils.py:55-66.
* images are read with Pillow and handed out in OpenCV's BGR channel order, because that is
  what ``cv.imread`` gives the reference.

The dataset itself (``../diff_1/raw_data``) is not distributed with the reference
(``.gitignore:1``); every path function therefore takes an optional ``root``.
"""
from __future__ import annotations

import os

import numpy as np

from . import _native

__all__ = ["get_base_path", "get_path", "imread", "equalize_hist", "visualize_equalized_hist", "get_no_scat_img",
           "get_features", "save2mat"]

DEFAULT_ROOT = "../diff_1/raw_data"     # utils.py:28


def get_base_path(case_idx: int = 1, root: str = DEFAULT_ROOT):
    """utils.py:27-28."""
    return f"{root}/case{case_idx}"


def get_path(case_idx: int = 1, img_idx: int = 1, no_scat=False, root: str = DEFAULT_ROOT):
    """utils.py:30-31: ``scat/img_haze{i}.png`` or ``no_scat/img_nohaze{i}.png``."""
    return (f"{get_base_path(case_idx, root)}/{'no_' if no_scat else ''}scat/"
            f"img_{'no' if no_scat else ''}haze{img_idx}.png")


def imread(path):
    """``cv.imread(path)``: (h, w, 3) uint8 in B, G, R order; ``None`` when the file cannot be
    read (OpenCV's convention, which the reference relies on implicitly)."""
    try:
        from PIL import Image
        with Image.open(path) as im:
            rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
    except (OSError, ValueError):
        return None
    return np.ascontiguousarray(rgb[..., ::-1])


def equalize_hist(img, device=-1):
    """``np.stack([cv.equalizeHist(img[..., i]) for i in range(c)], axis=-1)`` (utils.py:88),
    or ``cv.equalizeHist(img)`` for a single plane."""
    return _native.equalize_hist(img, device=device)


def visualize_equalized_hist(case_idx=1, img_idx=3, disp=False, root: str = DEFAULT_ROOT, device=-1):
    """utils.py:85-91.  ``disp`` is accepted for signature compatibility; there is no window
    system to show the image in."""
    path = get_path(case_idx, img_idx, root=root)
    img = imread(path)
    if img is None:
        raise FileNotFoundError(path)
    return equalize_hist(img, device=device)


def get_no_scat_img(case_idx, img_idx, center_id, root: str = DEFAULT_ROOT):
    """utils.py:50-53: ``(centre image, other image)`` of the haze-free set."""
    return (imread(get_path(case_idx, center_id, True, root=root)),
            imread(get_path(case_idx, img_idx, True, root=root)))


def get_features(case_idx: int = 1, pic_id: int = 1, center_id: int = 3, root: str = DEFAULT_ROOT):
    """utils.py:55-66: matched keypoints of (centre picture, picture ``pic_id``) from
    ``keypoints.mat``.  The file holds one 6 x n matrix per non-centre picture: rows 0-1 the
    centre picture's x, y, row 2 ones, rows 3-4 the other picture's x, y, row 5 ones."""
    import scipy.io
    feature_mat = scipy.io.loadmat(f"{get_base_path(case_idx, root)}/keypoints.mat")["keypoints"]
    valid_pic_id = {1, 2, 3, 4, 5}
    valid_pic_id.remove(center_id)
    if pic_id not in valid_pic_id:
        raise ValueError(f"{pic_id} not valid (3 is the id of the center image, therefore [1, 2, 4, 5] are available)")
    feat_mat = feature_mat[pic_id - 1 if pic_id < center_id else pic_id - 2][0]
    raw_kpts_op = feat_mat[3:-1, :].T
    raw_kpts_cp = feat_mat[:2, :].T
    return raw_kpts_cp, raw_kpts_op


def savemat_f64(file_name, name, arr):
    """``scipy.io.savemat(file_name, {name: arr})`` for ONE real float64 2-D array - the file the reference writes
    (apap.py:264, utils.py:68-70) - without importing scipy (70 ms of a 300 ms command): MAT 5, uncompressed, little endian;
    byte for byte what scipy writes (tests/test_host_logic.py compares them), the creation time in the header included."""
    import os as _os
    import struct
    import time
    arr = np.asarray(arr)
    nm = name.encode("latin1")
    if arr.dtype != np.float64 or arr.ndim != 2 or not 0 < len(nm) < 64:
        raise ValueError("savemat_f64 writes a 2-D float64 array under a short name")
    text = f"MATLAB 5.0 MAT-file Platform: {_os.name}, Created on: {time.asctime()}".encode("latin1")
    head = text[:116].ljust(116, b"\0") + struct.pack("<q", 0) + struct.pack("<H", 0x0100) + b"IM"

    def element(mdtype, payload):
        n = len(payload)
        if 0 < n <= 4:                                   # "small data element": type and size share one word
            return struct.pack("<HH", mdtype, n) + payload.ljust(4, b"\0")
        return struct.pack("<II", mdtype, n) + payload + b"\0" * ((-n) % 8)
    body = element(6, struct.pack("<II", 6, 0))                       # array flags: mxDOUBLE_CLASS, nzmax 0
    body += element(5, struct.pack("<ii", *arr.shape))                # dimensions
    body += element(1, nm)                                            # array name
    body += element(9, np.asfortranarray(arr).tobytes(order="F"))     # real part, column-major
    with open(file_name, "wb") as fh:
        fh.write(head + struct.pack("<II", 14, len(body)) + body)      # miMATRIX


def save2mat(path: str, arr, name: str = "sift_feature", prefix: str = "./output/"):
    """utils.py:68-70.  The reference's one use on this path - a float64 (m*m, 9) array under 'H' - takes the writer above;
    anything else goes to scipy."""
    a = np.asarray(arr)
    if a.dtype == np.float64 and a.ndim == 2 and a.size > 0:
        return savemat_f64(f"{prefix}{path}.mat", name, a)
    import scipy.io
    scipy.io.savemat(f"{prefix}{path}.mat", {name: arr})
