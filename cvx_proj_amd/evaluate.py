"""RMSE evaluator for the ``.mat`` files the APAP path writes (second half).

The reference checks its results with a MATLAB script that is not in its tree
(``diff_1/program/main_example.m``, README.md:66-68,83,153-160; called from
``pyviz/grid_search.sh:57-59``): it loads the homography from the ``.mat``, projects
ground-truth points and prints the RMSE - the numbers quoted in ``pyviz/results/case_*.config``
(``APAP RMSE haze: 4.0591``, ``case_1.config:2``).  This module restates that check in Python for
the file layout of ``apap.py:250-265``:

* key ``H``, shape ``(m*m, 9)`` float64; row ``i*m + j`` is the COLUMN-major flattening of the
  inverted, ``[2,2]``-normalised homography of mesh cell ``(i, j)`` - it maps centre-image
  coordinates to other-image coordinates (``apap.py:250-254,263-264``);
* a ``(3, 3)`` ``H`` (what ``spectral_method.py:241`` and the RANSAC baseline write) is one global
  matrix for the whole canvas.

A ground-truth point of the centre image lies in the mesh cell its canvas pixel
``(x + off_x, y + off_y)`` falls into - the cell ``local_warp`` would use for that pixel
(``apap.py:207-210``: first edge above the index, minus one) - and is projected through that
cell's matrix.  Host-side numpy: a few hundred points per pair.

    python -m cvx_proj_amd.evaluate [case_idx] [img_idx] --gt pts.mat
        (--geometry FW,FH,OX,OY | --synth C1 | --pair pair.npz) [--in-prefix ../diff_1/results/]
"""
from __future__ import annotations

import sys

import numpy as np

from .geometry import final_size, get_mesh

__all__ = ["cells_of_points", "project_through_grid", "rmse", "load_h_mat", "load_ground_truth", "main"]


def load_h_mat(path):
    """``(grid, m)``: ``grid`` float64 ``(m, m, 3, 3)`` of centre -> other matrices (``m = 1`` for a
    global 3 x 3 file)."""
    import scipy.io
    H = np.asarray(scipy.io.loadmat(path)["H"], dtype=np.float64)
    if H.shape == (3, 3):
        return H.reshape(1, 1, 3, 3), 1
    if H.ndim != 2 or H.shape[1] != 9:
        raise ValueError(f"{path}: key 'H' has shape {H.shape}; expected (m*m, 9) or (3, 3)")
    m = int(round(np.sqrt(H.shape[0])))
    if m * m != H.shape[0]:
        raise ValueError(f"{path}: {H.shape[0]} rows is not a square mesh")
    # rows are column-major 3 x 3 matrices: undo the transpose of apap.py:263
    return H.reshape(m, m, 3, 3).transpose(0, 1, 3, 2), m


def load_ground_truth(path):
    """``(centre (k, 2), other (k, 2))`` float64 from a ``.mat``/``.npz`` holding either the keys
    ``center`` and ``other`` (k x 2 each) or one 6 x k matrix in the layout of the reference's
    ``keypoints.mat`` entries (rows 0-1 centre x, y; rows 3-4 other x, y - ``utils.py:63-66``)."""
    if path.endswith(".npz"):
        z = dict(np.load(path))
    else:
        import scipy.io
        z = {k: v for k, v in scipy.io.loadmat(path).items() if not k.startswith("__")}
    if "center" in z and "other" in z:
        c, o = np.asarray(z["center"], dtype=np.float64), np.asarray(z["other"], dtype=np.float64)
    else:
        mats = [v for v in z.values() if isinstance(v, np.ndarray) and v.ndim == 2 and v.shape[0] == 6]
        if len(mats) != 1:
            raise ValueError(f"{path}: need keys 'center' and 'other', or exactly one 6 x k matrix")
        c, o = mats[0][0:2].T.astype(np.float64), mats[0][3:5].T.astype(np.float64)
    if c.shape != o.shape or c.ndim != 2 or c.shape[1] != 2:
        raise ValueError(f"{path}: point arrays must both be (k, 2); got {c.shape} and {o.shape}")
    return c, o


def cells_of_points(points, mesh, offsets):
    """Mesh cell ``(row, col)`` of every centre-image point: the cell ``local_warp`` uses for the
    canvas pixel the point falls into (``apap.py:207-210``).  ``mesh`` is ``get_mesh``'s ``(2, m + 1)``
    edge array ``[mesh_w; mesh_h]``; a point outside the canvas raises IndexError like the
    reference's ``np.where(...)[0][0]``."""
    mesh_w, mesh_h = np.asarray(mesh[0], dtype=np.float64), np.asarray(mesh[1], dtype=np.float64)
    pts = np.asarray(points, dtype=np.float64)
    px = np.floor(pts[:, 0] + offsets[0])        # the pixel that contains the point
    py = np.floor(pts[:, 1] + offsets[1])

    def lookup(idx, edges):
        above = idx[:, None] < edges[None, :]
        if not above.any(axis=1).all():
            raise IndexError("index 0 is out of bounds for axis 0 with size 0 (point outside the mesh)")
        return np.argmax(above, axis=1) - 1      # -1 wraps to the last cell, as a Python index does

    return lookup(py, mesh_h), lookup(px, mesh_w)


def project_through_grid(grid, mesh, offsets, points):
    """Other-image positions of centre-image ``points`` through the per-cell centre -> other
    matrices ``grid`` ``(m_r, m_c, 3, 3)``."""
    pts = np.asarray(points, dtype=np.float64)
    if grid.shape[:2] == (1, 1):
        hc = np.broadcast_to(grid[0, 0], (len(pts), 3, 3))
    else:
        r, c = cells_of_points(pts, mesh, offsets)
        hc = grid[r, c]
    homog = np.concatenate([pts, np.ones((len(pts), 1))], axis=1)
    t = np.einsum("kij,kj->ki", hc, homog)
    return t[:, :2] / t[:, 2:3]


def rmse(grid, mesh, offsets, gt_center, gt_other):
    """Root of the mean squared distance between the projected centre points and their ground-truth
    positions in the other image, in pixels."""
    d = project_through_grid(grid, mesh, offsets, gt_center) - np.asarray(gt_other, dtype=np.float64)
    return float(np.sqrt(np.mean(np.sum(d * d, axis=1))))


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(prog="cvx_proj_amd.evaluate", description=__doc__.split("\n\n")[0])
    ap.add_argument("case_idx", nargs="?", type=int, default=1)
    ap.add_argument("img_idx", nargs="?", type=int, default=1)
    ap.add_argument("--gt", required=True, help=".mat / .npz with the ground-truth point pairs (see load_ground_truth)")
    ap.add_argument("--in-prefix", default="../diff_1/results/", help="where apap.py wrote case{c}/H3{i}_apap.mat")
    ap.add_argument("--suffix", default="_apap", help="file is H3{i}{suffix}.mat")
    ap.add_argument("--geometry", help="canvas FW,FH,OX,OY (what final_size returned for the pair)")
    ap.add_argument("--synth", help="take the canvas of this synthetic configuration of cvx_proj_amd.synth")
    ap.add_argument("--pair", help=".npz with H (3,3), other_shape, center_shape: the canvas is final_size of them")
    a = ap.parse_args(argv)
    grid, m = load_h_mat(f"{a.in_prefix}case{a.case_idx}/H3{a.img_idx}{a.suffix}.mat")
    if a.geometry:
        fw, fh, ox, oy = (int(v) for v in a.geometry.split(","))
    elif a.synth:
        from .synth import config_pair
        p = config_pair(a.synth, with_image=False)
        fw, fh, ox, oy = p.final_w, p.final_h, p.off_x, p.off_y
    elif a.pair:
        z = np.load(a.pair)

        class _S:
            def __init__(self, shape):
                self.shape = tuple(shape)
        fw, fh, ox, oy = (int(v) for v in final_size(_S(z["center_shape"]), _S(z["other_shape"]), z["H"]))
    elif m == 1:
        fw = fh = 1
        ox = oy = 0
    else:
        ap.error("a per-cell grid needs the canvas: give --geometry, --synth or --pair")
    mesh = get_mesh((fw, fh), m + 1)
    c, o = load_ground_truth(a.gt)
    value = rmse(grid, mesh, (ox, oy), c, o)
    print(f"APAP RMSE: {value:.4f}  ({len(c)} points, {m}x{m} cells, canvas {fw}x{fh}, offsets ({ox},{oy}))")
    return 0


if __name__ == "__main__":
    sys.exit(main())
