"""Canvas / mesh geometry for the APAP engine (host side, numpy only).

These are the small set-up helpers the reference keeps in ``pyviz/apap_utils.py``;
they run once per image pair and produce a few kilobytes, so they stay on the host.
Each function documents the reference lines whose results it reproduces.
"""
from __future__ import annotations

import numpy as np


def get_mesh(size, mesh_size, start=0):
    """Edges of the warp cells: ``mesh_size`` evenly spaced values from ``start`` to
    the canvas width (row 0) and height (row 1).  Callers pass ``cells + 1``.

    Same values as reference ``apap_utils.py:10-21``; returns float64 ``(2, mesh_size)``.
    """
    width, height = size
    edges = np.empty((2, mesh_size), dtype=np.float64)
    edges[0] = np.linspace(start, width, mesh_size)
    edges[1] = np.linspace(start, height, mesh_size)
    return edges


def get_vertice(size, mesh_size, offsets):
    """Sample point of every mesh cell, in source-image coordinates.

    ``out[i, j] = (x_j - offset_x, y_i - offset_y)`` with ``x = linspace(0, w, m) +
    w / (2 m)`` - note the spacing ``w / (m - 1)``: these are not the centres of the
    cells of :func:`get_mesh`.  That is what reference ``apap_utils.py:23-38`` computes
    and it is kept.  Returns float64 ``(m, m, 2)``.
    """
    width, height = size
    col_x = np.linspace(0, width, mesh_size) + width / (mesh_size * 2)
    row_y = np.linspace(0, height, mesh_size) + height / (mesh_size * 2)
    out = np.empty((mesh_size, mesh_size, 2), dtype=np.float64)
    # (x - offset) per column / row, then broadcast: the same float64 subtraction the reference applies to the
    # whole (m, m, 2) array, one pass over it instead of three
    off = np.array(offsets)
    out[..., 0] = (col_x - off[0])[None, :]
    out[..., 1] = (row_y - off[1])[:, None]
    return out


def final_size(src_img, dst_img, project_H):
    """Canvas that holds ``dst_img`` and ``src_img`` warped by ``project_H``.

    Returns ``(width, height, offset_x, offset_y)`` as reference ``apap_utils.py:40-73``
    does: the four source corners ``(0,0) (0,h) (w,0) (w,h)`` are projected with
    float32 points, truncated toward zero, and united with the destination rectangle.
    (The reference spells the truncation ``astype(np.int)``, which modern numpy no longer
    has; ``int`` is the same type.)
    """
    src_h, src_w = src_img.shape[:2]
    pts = np.float32([[0, 0, 1], [0, src_h, 1], [src_w, 0, 1], [src_w, src_h, 1]])
    xs, ys = [], []
    for pt in pts:
        v = np.matmul(project_H, pt)
        xs.append(v[0] / v[2])
        ys.append(v[1] / v[2])
    xs = np.array(xs).astype(int)
    ys = np.array(ys).astype(int)
    dst_h, dst_w = dst_img.shape[:2]
    min_x, max_x = min(xs.min(), 0), max(xs.max(), dst_w)
    min_y, max_y = min(ys.min(), 0), max(ys.max(), dst_h)
    return (max_x - min_x, max_y - min_y,
            -min_x if min_x < 0 else 0, -min_y if min_y < 0 else 0)


def uniform_blend(img1, img2):
    """Average two canvases where both are non-black, add them elsewhere.

    A pixel counts as non-black when its channel mean is > 0.  Sum in float64, halve
    in the overlap, cast to uint8 - reference ``apap_utils.py:75-88``.
    """
    both = (img1.mean(axis=-1) > 0) & (img2.mean(axis=-1) > 0)
    total = img1.astype(np.float64) + img2.astype(np.float64)
    total *= np.where(both, 0.5, 1.0)[..., None]
    return total.astype(np.uint8)
