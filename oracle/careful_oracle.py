"""Specification of the engine's CAREFUL PATH in numpy (TEST INFRASTRUCTURE, not product).

Where the weighted DLT system is numerically rank-deficient the engine does not solve the normal
equations ``A^T W^2 A`` but works on the weighted rows themselves, like the reference's SVD of ``W A``
(``apap.py:159-161``) - by a route chosen for a GPU lane: row-insertion Givens QR into a 9 x 9
triangle, then one-sided Jacobi on the columns of ``R^T`` (``qr_resolve`` in
``cvx_proj_amd/csrc/apap_kernels.hip``).  This module restates that route step by step so that the CPU
suite can check the ALGORITHM against the golden vectors (``tests/golden/illcond_*.npz``: the
reference's grids and 60-digit exact answers) without a GPU; the GPU suite checks the kernel.
Only ``tests/`` may import it.
"""
from __future__ import annotations

import numpy as np

from . import apap_oracle as O

__all__ = ["givens_qr_r", "jacobi_columns", "smallest_right_singular_vector", "local_homography_careful"]


def givens_qr_r(A):
    """Upper-triangular ``R`` (9 x 9) with ``R^T R = A^T A``: rows of ``A`` rotated in one by one."""
    R = np.zeros((9, 9))
    for row in np.asarray(A, dtype=np.float64):
        r = row.copy()
        for k in range(9):
            if r[k] == 0.0:
                continue
            a, b = R[k, k], r[k]
            h = np.hypot(a, b)                    # the kernel scales by a power of two first: same value
            c, s = a / h, b / h
            Rk = R[k, k:].copy()
            R[k, k:] = c * Rk + s * r[k:]
            r[k:] = c * r[k:] - s * Rk
    return R


def jacobi_columns(G, sweeps=30, tol=1e-15):
    """One-sided (Hestenes) Jacobi: rotate column pairs until all are orthogonal.  Returns the rotated
    matrix; its columns are ``sigma_j u_j`` of the input."""
    G = np.array(G, dtype=np.float64)
    n = G.shape[1]
    for _ in range(sweeps):
        rotated = False
        for p in range(n - 1):
            for q in range(p + 1, n):
                alpha, beta, gam = G[:, p] @ G[:, p], G[:, q] @ G[:, q], G[:, p] @ G[:, q]
                if not abs(gam) > tol * np.sqrt(alpha * beta):
                    continue
                rotated = True
                zeta = (beta - alpha) / (2.0 * gam)
                t = np.copysign(1.0, zeta) / (abs(zeta) + np.sqrt(zeta * zeta + 1.0))
                c = 1.0 / np.sqrt(t * t + 1.0)
                s = c * t
                gp = G[:, p].copy()
                G[:, p] = c * gp - s * G[:, q]
                G[:, q] = s * gp + c * G[:, q]
        if not rotated:
            break
    return G


def smallest_right_singular_vector(A, pick_rank=0):
    """Last row of the thin ``V^T`` of ``A`` (2n x 9): the column of ``R^T W`` with the
    ``pick_rank``-th smallest norm (``pick_rank`` = 9 - min(2n, 9) exact zeros sort below it)."""
    R = givens_qr_r(A)
    big = np.abs(R).max()
    G = jacobi_columns(R.T / (2.0 ** np.frexp(big)[1]) if big > 0 else R.T)
    nrm2 = np.sum(G * G, axis=0)
    # descending singular values, equal ones in index order: rank = how many sort after this column
    rank = [sum(1 for j in range(9) if j != i and (nrm2[j] < nrm2[i] or (nrm2[j] == nrm2[i] and j > i))) for i in range(9)]
    best = rank.index(pick_rank)
    if nrm2[best] > 0:
        return G[:, best] / np.sqrt(nrm2[best])
    e = np.zeros(9)
    e[best] = 1.0
    return e


def local_homography_careful(src_point, dst_point, vertices, gamma, sigma, cells=None):
    """``local_homography`` with EVERY cell on the careful path (the engine takes it only where its
    conditioning guard fires).  float32 ``(rows, cols, 3, 3)``."""
    n = src_point.shape[0]
    rows, cols, _ = vertices.shape
    p = O.prepare(src_point, dst_point)
    aa = p["aa"].astype(np.float64)
    pick = 9 - min(2 * n, 9)
    H = np.zeros((rows, cols, 3, 3), dtype=np.float32)
    for i, j in (cells if cells is not None else ((i, j) for i in range(rows) for j in range(cols))):
        w = O.cell_weights(vertices[i, j], src_point, gamma, sigma)
        H[i, j] = O._denormalise(smallest_right_singular_vector(np.repeat(w, 2)[:, None] * aa, pick), p)
    return H
