"""CPU oracle for the APAP moving-DLT hot path (TEST INFRASTRUCTURE, not product).

This module is a numpy restatement of the algorithm in the reference's
``pyviz/apap.py`` and ``pyviz/apap_utils.py`` (Enigmatisms/cvx_proj).  It exists so
that the HIP engine in ``cvx_proj_amd`` has something to be compared with on
machines where the reference itself is not present (the GPU box).

Rules of use
------------
* Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
  ``bench.py`` may import this module, and only as the checker / the reported CPU
  baseline.  Nothing under ``cvx_proj_amd/`` imports it; the product path fails
  loudly when the HIP library is missing.
* Pinning: ``tests/golden/*.npz`` hold outputs of the *reference itself*, produced in
  the build container by ``tests/golden/make_golden.py`` (which imports
  ``/root/reference/pyviz/apap.py`` in place).  ``tests/test_oracle_golden.py`` checks
  every function here against those vectors.
* One third-party call on the path is absent from both the reference tree and this
  image: ``cv2.SVDecomp`` (opencv-python 4.6.0.66, reference ``apap.py:160``).  The
  golden vectors were produced with ``numpy.linalg.svd`` serving that one call, and
  the oracle uses the same.  PARITY IS THEREFORE UNPINNED AT THE BIT LEVEL OF
  ``cv::SVDecomp``; it is pinned mathematically (the singular vector of the smallest
  singular value is unique up to sign, the sign is removed by ``h / h[2, 2]``, and
  the result is stored as float32).

Two flavours are provided for the two hot loops:
* ``*_loop``  - the reference's loop structure, temporaries and dtypes, one mesh
  cell / one pixel per Python iteration.  This is what ``bench.py`` times as the
  CPU baseline ("port").
* ``*_fast``  - vectorised numpy (normal equations + ``eigh``; vectorised gather)
  used to check large configurations in seconds.  Checked against ``*_loop``.
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "get_mesh", "get_vertice", "final_size", "uniform_blend",
    "normalize_2d_pts", "conditioner_from_pts", "point_normalize", "dlt_rows",
    "prepare", "cell_weights", "local_homography_loop", "local_homography_fast", "local_homography_pool",
    "local_homography_exact_cell",
    "invert_cells_f32", "cell_lookup", "local_warp_loop", "local_warp_fast",
    "warp_coords_fast", "stitch", "invert_normalize_flatten", "project", "reprojection_rmse_delta",
]


# --------------------------------------------------------------------------------------
# apap_utils.py helpers
# --------------------------------------------------------------------------------------
def get_mesh(size, mesh_size, start=0):
    """Cell edges along x and y.  Reference: apap_utils.py:10-21."""
    w, h = size
    return np.stack([np.linspace(start, w, mesh_size), np.linspace(start, h, mesh_size)], axis=0)


def get_vertice(size, mesh_size, offsets):
    """Per-cell sample points, ``[i, j] = (x_j, y_i) - offsets``.  Reference:
    apap_utils.py:23-38.  Spacing is ``w / (mesh_size - 1)`` shifted by
    ``w / (2 mesh_size)`` - reproduced as-is."""
    w, h = size
    xs = np.linspace(0, w, mesh_size) + w / (mesh_size * 2)
    ys = np.linspace(0, h, mesh_size) + h / (mesh_size * 2)
    gx, gy = np.meshgrid(xs, ys)
    v = np.stack([gx, gy], axis=-1)
    v -= np.array(offsets)
    return v


def final_size(src_img, dst_img, project_H):
    """Canvas size and offsets.  Reference: apap_utils.py:40-73 (``np.int`` there is
    the builtin ``int``: truncation toward zero)."""
    h, w = src_img.shape[:2]
    corners = []
    for pt in (np.float32([0, 0, 1]), np.float32([0, h, 1]), np.float32([w, 0, 1]), np.float32([w, h, 1])):
        vec = np.matmul(project_H, pt)
        corners.append([vec[0] / vec[2], vec[1] / vec[2]])
    corners = np.array(corners).astype(int)
    h, w = dst_img.shape[:2]
    max_x = max(np.max(corners[:, 0]), w)
    max_y = max(np.max(corners[:, 1]), h)
    min_x = min(np.min(corners[:, 0]), 0)
    min_y = min(np.min(corners[:, 1]), 0)
    width = max_x - min_x
    height = max_y - min_y
    offset_x = -min_x if min_x < 0 else 0
    offset_y = -min_y if min_y < 0 else 0
    return width, height, offset_x, offset_y


def uniform_blend(img1, img2):
    """Reference: apap_utils.py:75-88."""
    g = (np.mean(img1, axis=-1) > 0) & (np.mean(img2, axis=-1) > 0)
    out = img1.astype(np.float64) + img2.astype(np.float64)
    mask = np.where(g, 0.5, 1.0)[..., None]
    return (out * mask).astype(np.uint8)


# --------------------------------------------------------------------------------------
# host-side preparation (apap.py:35-119)
# --------------------------------------------------------------------------------------
def normalize_2d_pts(point):
    """Hartley similarity: centroid to 0, mean distance to sqrt(2).  Reference:
    apap.py:35-59.  float32 in -> (t float32 3x3, transformed points float32 (N,2))."""
    n = point.shape[0]
    c = np.mean(point, axis=0)
    pt = point - c
    mean_dist = np.mean(np.sqrt(np.sum(np.square(pt), axis=1)))
    scale = np.sqrt(2) / (mean_dist + 1e-8)
    t = np.array([[scale, 0, -scale * c[0]], [0, scale, -scale * c[1]], [0, 0, 1]], dtype=np.float32)
    homog = np.column_stack((point.copy(), np.ones(n, dtype=np.float32)))
    new_point = t.dot(homog.T).T[:, :2]
    return t, new_point


def conditioner_from_pts(point):
    """Per-axis conditioner with the sample standard deviation.  Reference:
    apap.py:63-89."""
    n = point.shape[0]
    mean_x, mean_y = np.mean(point, axis=0)
    std = np.std(point, axis=0)
    std = np.sqrt(std * std * n / (n - 1))
    std_x, std_y = std
    std_x = std_x + (std_x == 0)
    std_y = std_y + (std_y == 0)
    norm_x = np.sqrt(2) / std_x
    norm_y = np.sqrt(2) / std_y
    return np.array([[norm_x, 0, -norm_x * mean_x], [0, norm_y, -norm_y * mean_y], [0, 0, 1]], dtype=np.float32)


def point_normalize(nf, c):
    """Reference: apap.py:92-100 (a Python loop there; the same float32 arithmetic
    done array-wise here)."""
    cf = np.zeros_like(nf)
    cf[:, 0] = nf[:, 0] * c[0, 0] + c[0, 2]
    cf[:, 1] = nf[:, 1] * c[1, 1] + c[1, 2]
    return cf


def dlt_rows(cf1, cf2):
    """The 2N x 9 DLT matrix, float32 with float32-rounded products.  Reference:
    apap.py:103-119."""
    n = cf1.shape[0]
    a = np.zeros((2 * n, 9), dtype=np.float32)
    x, y = cf1[:, 0], cf1[:, 1]
    xp, yp = cf2[:, 0], cf2[:, 1]
    a[0::2, 0] = x
    a[0::2, 1] = y
    a[0::2, 2] = 1
    a[0::2, 6] = (-xp) * x
    a[0::2, 7] = (-xp) * y
    a[0::2, 8] = -xp
    a[1::2, 3] = x
    a[1::2, 4] = y
    a[1::2, 5] = 1
    a[1::2, 6] = (-yp) * x
    a[1::2, 7] = (-yp) * y
    a[1::2, 8] = -yp
    return a


def prepare(src_point, dst_point):
    """Everything ``local_homography`` computes once before its cell loop
    (apap.py:132-145).  Returns a dict."""
    N1, nf1 = normalize_2d_pts(src_point)
    N2, nf2 = normalize_2d_pts(dst_point)
    C1 = conditioner_from_pts(nf1)
    C2 = conditioner_from_pts(nf2)
    cf1 = point_normalize(nf1, C1)
    cf2 = point_normalize(nf2, C2)
    aa = dlt_rows(cf1, cf2)
    return dict(N1=N1, N2=N2, C1=C1, C2=C2, nf1=nf1, nf2=nf2, cf1=cf1, cf2=cf2, aa=aa,
                iC2=np.linalg.inv(C2), iN2=np.linalg.inv(N2))


# --------------------------------------------------------------------------------------
# hot loop 1: per-cell weighted DLT (apap.py:147-168)
# --------------------------------------------------------------------------------------
def cell_weights(vertex, src_point, gamma, sigma):
    """``w_k = max(exp(-|v - src_k| / sigma^2), gamma)``, float64.  Reference:
    apap.py:142,150-152."""
    inverse_sigma = 1.0 / (sigma ** 2)
    dist = np.tile(vertex, (src_point.shape[0], 1)) - src_point
    weight = np.exp(-(np.sqrt(dist[:, 0] ** 2 + dist[:, 1] ** 2) * inverse_sigma))
    weight[weight < gamma] = gamma
    return weight


def _denormalise(h, p):
    """apap.py:163-167: h <- inv(C2) h C1; h <- inv(N2) h N1; h /= h[2,2]."""
    h = h.reshape(3, 3)
    h = p["iC2"].dot(h).dot(p["C1"])
    h = p["iN2"].dot(h).dot(p["N1"])
    return h / h[2, 2]


def local_homography_loop(src_point, dst_point, vertices, gamma, sigma, cells=None, want_weights=True, cond_out=None):
    """Faithful restatement of ``APAP.local_homography`` (apap.py:121-169): one
    weighted 2N x 9 SVD per mesh cell in float64, float32 store.  ``numpy.linalg.svd``
    stands in for ``cv.SVDecomp``.  ``cells`` optionally restricts the loop to a list
    of (i, j) pairs (used by the bounded CPU-baseline timing); other cells stay 0.

    ``cond_out`` (a float64 array of the mesh's shape, test diagnostics only) receives per cell
    ``sigma_1 / (sigma_k-1 - sigma_k)`` of the weighted system, k = the index of the vector taken:
    times the machine epsilon it is the forward error of that singular vector in ANY backward-stable
    float64 SVD - how far the reference's own float64 result is from the exact one."""
    n = src_point.shape[0]
    rows, cols, _ = vertices.shape
    p = prepare(src_point, dst_point)
    aa = p["aa"]
    H = np.zeros((rows, cols, 3, 3), dtype=np.float32)
    W = np.zeros((rows, cols, n)) if want_weights else None
    it = cells if cells is not None else ((i, j) for i in range(rows) for j in range(cols))
    for i, j in it:
        weight = cell_weights(vertices[i, j], src_point, gamma, sigma)
        if want_weights:
            W[i, j, :] = weight
        A = np.expand_dims(np.repeat(weight, 2), -1) * aa
        _, sv, vt = np.linalg.svd(A, full_matrices=False)
        H[i, j] = _denormalise(vt[-1, :], p)
        if cond_out is not None:
            gap = (sv[-2] - sv[-1]) if len(sv) > 1 else sv[-1]
            cond_out[i, j] = sv[0] / gap if gap > 0 else np.inf
    return H, W


def local_homography_exact_cell(src_point, dst_point, vertex, gamma, sigma, digits=60, gram=True):
    """The reference's answer for ONE cell with its SVD taken in ``digits``-digit arithmetic (mpmath): the
    same float64 matrix ``repeat(weight, 2)[:, None] * aa`` (apap.py:150-159), the last right singular
    vector of its thin SVD, de-normalised with the reference's float32 matrices, float32 result.  The
    arbiter where the engine and the reference's float64 LAPACK SVD disagree (test diagnostics only;
    ~50 ms per cell)."""
    import mpmath as mp
    p = prepare(src_point, dst_point)
    A = np.expand_dims(np.repeat(cell_weights(vertex, src_point, gamma, sigma), 2), -1) * p["aa"]
    with mp.workdps(digits):
        if A.shape[0] <= 2000 or not gram:
            _, S, V = mp.svd_r(mp.matrix(A.tolist()), full_matrices=False, compute_uv=True)
            k = min(range(len(S)), key=lambda i: S[i])
            v = np.array([float(V[k, j]) for j in range(9)])
        else:
            # thousands of rows: the right singular vectors as eigenvectors of the EXACTLY accumulated A^T A (integer arithmetic on
            # the float64 entries' exact values), 9 x 9 in `digits` digits - the same vector to ~digits / 2 - 16 places for these
            # systems (sigma_1 / (sigma_8 - sigma_9) ~ 1e3-1e5) at a hundredth of the cost of a 10 000 x 9 multiprecision SVD
            from fractions import Fraction
            G = [[Fraction(0)] * 9 for _ in range(9)]
            rows = [[Fraction(float(x)) for x in r] for r in A]
            for r in rows:
                for i in range(9):
                    if r[i]:
                        for j in range(i, 9):
                            G[i][j] += r[i] * r[j]
            M = mp.matrix(9, 9)
            for i in range(9):
                for j in range(i, 9):
                    M[i, j] = M[j, i] = mp.mpf(G[i][j].numerator) / mp.mpf(G[i][j].denominator)
            E, Q = mp.eigsy(M)
            k = min(range(9), key=lambda i: E[i])
            v = np.array([float(Q[j, k]) for j in range(9)])
    return _denormalise(v, p).astype(np.float32)


def _pool_worker(args):
    """One worker of ``local_homography_pool``: the faithful loop over a slice of cells."""
    src, dst, vertices, gamma, sigma, cells = args
    H, _ = local_homography_loop(src, dst, vertices, gamma, sigma, cells=cells, want_weights=False)
    return [(i, j, H[i, j]) for i, j in cells]


def _pool_init(counter):
    with counter.get_lock():
        counter.value += 1


def local_homography_pool(src_point, dst_point, vertices, gamma, sigma, cells, workers):
    """"Best-effort CPU" row of BASELINE.md: the faithful per-cell loop spread over a process
    pool (one BLAS thread per worker).  Used only by bench.py's cpu_baseline leg.  Returns
    ``(H, seconds)``; the seconds exclude worker start-up (interpreter + imports)."""
    import multiprocessing as mp
    import time
    chunks = [cells[k::workers] for k in range(workers)]
    ctx = mp.get_context("spawn")      # never fork a process that has initialised the GPU
    ready = ctx.Value("i", 0)
    with ctx.Pool(workers, initializer=_pool_init, initargs=(ready,)) as pool:
        deadline = time.perf_counter() + 60.0
        while ready.value < workers:   # every worker has started and imported this module (numpy with it)
            if time.perf_counter() > deadline:      # a worker died in its import or initializer: do not hang the caller
                raise RuntimeError(f"local_homography_pool: only {ready.value} of {workers} workers started within 60 s")
            time.sleep(0.01)
        t0 = time.perf_counter()
        parts = pool.map(_pool_worker, [(src_point, dst_point, vertices, gamma, sigma, c) for c in chunks if c])
        seconds = time.perf_counter() - t0
    H = np.zeros(vertices.shape[:2] + (3, 3), dtype=np.float32)
    for part in parts:
        for i, j, h in part:
            H[i, j] = h
    return H, seconds


def moments_from_rows(aa):
    """Per-point 9x9 outer-product sums r1 r1^T + r2 r2^T in float64 (products of two
    float32 values are exact in float64).  Returns (N, 9, 9)."""
    a = aa.astype(np.float64)
    r1, r2 = a[0::2], a[1::2]
    return r1[:, :, None] * r1[:, None, :] + r2[:, :, None] * r2[:, None, :]


def local_homography_fast(src_point, dst_point, vertices, gamma, sigma, chunk=4096, want_weights=False):
    """Vectorised restatement: ``A^T diag(w^2) A`` by a matrix product over points,
    smallest eigenvector by ``numpy.linalg.eigh``.  Mathematically the right singular
    vector the loop version takes; checked against it in tests."""
    n = src_point.shape[0]
    rows, cols, _ = vertices.shape
    p = prepare(src_point, dst_point)
    P = moments_from_rows(p["aa"]).reshape(n, 81)
    inverse_sigma = 1.0 / (sigma ** 2)
    v = vertices.reshape(-1, 2)
    s = src_point.astype(np.float64)
    H = np.zeros((rows * cols, 3, 3), dtype=np.float32)
    W = np.zeros((rows * cols, n)) if want_weights else None
    iC2, C1, iN2, N1 = (p[k].astype(np.float64) for k in ("iC2", "C1", "iN2", "N1"))
    for lo in range(0, v.shape[0], chunk):
        vv = v[lo:lo + chunk]
        dx = vv[:, None, 0] - s[None, :, 0]
        dy = vv[:, None, 1] - s[None, :, 1]
        w = np.exp(-(np.sqrt(dx ** 2 + dy ** 2) * inverse_sigma))
        w[w < gamma] = gamma
        if want_weights:
            W[lo:lo + chunk] = w
        M = ((w * w) @ P).reshape(-1, 9, 9)
        _, vec = np.linalg.eigh(M)
        # last row of the THIN V^T (apap.py:160-161): smallest eigenvalue when 2n >= 9,
        # otherwise the smallest of the 2n that the thin SVD keeps
        h = vec[:, :, 9 - min(2 * n, 9)].reshape(-1, 3, 3)
        h = iC2 @ h @ C1
        h = iN2 @ h @ N1
        h = h / h[:, 2:3, 2:3]
        H[lo:lo + chunk] = h
    H = H.reshape(rows, cols, 3, 3)
    if want_weights:
        W = W.reshape(rows, cols, n)
    return H, W


# --------------------------------------------------------------------------------------
# hot loop 2: backward warp (apap.py:186-217)
# --------------------------------------------------------------------------------------
def invert_cells_f32(local_h):
    """Per-cell float32 inverse, as the in-place loop at apap.py:201-203 does
    (``numpy.linalg.inv`` on a float32 3x3 -> LAPACK sgesv)."""
    out = np.empty_like(local_h)
    rows, cols = local_h.shape[:2]
    for i in range(rows):
        for j in range(cols):
            out[i, j] = np.linalg.inv(local_h[i, j])
    return out


def cell_lookup(count, edges):
    """For every integer index ``0 <= i < count``: ``first k with i < edges[k]``,
    minus one (apap.py:207,209-210).  -1 wraps to the last cell as Python indexing
    does.  Raises IndexError when no edge exceeds some index, as the reference."""
    idx = np.arange(count)
    lt = idx[:, None] < np.asarray(edges)[None, :]
    if not lt.any(axis=1).all():
        raise IndexError("index 0 is out of bounds for axis 0 with size 0")
    return np.argmax(lt, axis=1) - 1


def local_warp_loop(ori_img, local_h, mesh, final_wh, offset, rows_subset=None):
    """Faithful restatement of ``APAP.local_warp`` (apap.py:186-217): one output pixel
    per Python iteration.  ``local_h`` is inverted IN PLACE like the reference does.
    ``rows_subset`` restricts the pixel loop to some canvas rows (bounded timing)."""
    mesh_w, mesh_h = mesh
    ori_h, ori_w, _ = ori_img.shape
    final_w, final_h = final_wh
    off_x, off_y = offset
    warped = np.zeros([final_h, final_w, 3], dtype=np.uint8)
    rows, cols = local_h.shape[:2]
    for i in range(rows):
        for j in range(cols):
            local_h[i, j, :] = np.linalg.inv(local_h[i, j, :])
    for i in (range(final_h) if rows_subset is None else rows_subset):
        m = np.where(i < mesh_h)[0][0]
        for j in range(final_w):
            n = np.where(j < mesh_w)[0][0]
            hom = local_h[m - 1, n - 1, :]
            t = hom @ np.array([j - off_x, i - off_y, 1])
            t /= t[2]
            if 0 < t[0] < ori_w and 0 < t[1] < ori_h:
                warped[i, j, :] = ori_img[int(t[1]), int(t[0]), :]
    return warped


def warp_coords_fast(hinv, mesh, final_wh, offset):
    """Vectorised target coordinates (float64) of every canvas pixel given the
    ALREADY INVERTED float32 per-cell matrices.  Returns (tx, ty) of shape
    (final_h, final_w)."""
    mesh_w, mesh_h = mesh
    final_w, final_h = final_wh
    off_x, off_y = offset
    rc = cell_lookup(final_h, mesh_h)
    cc = cell_lookup(final_w, mesh_w)
    hc = hinv[rc][:, cc].astype(np.float64)          # (final_h, final_w, 3, 3)
    x = (np.arange(final_w) - off_x).astype(np.float64)[None, :]
    y = (np.arange(final_h) - off_y).astype(np.float64)[:, None]
    t0 = hc[..., 0, 0] * x + hc[..., 0, 1] * y + hc[..., 0, 2]
    t1 = hc[..., 1, 0] * x + hc[..., 1, 1] * y + hc[..., 1, 2]
    t2 = hc[..., 2, 0] * x + hc[..., 2, 1] * y + hc[..., 2, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        return t0 / t2, t1 / t2


def local_warp_fast(ori_img, hinv, mesh, final_wh, offset, band=256, rows=None):
    """Vectorised restatement of the pixel loop given ALREADY INVERTED cells.  Strict
    ``0 < t < size`` test and truncation as apap.py:214-215.  Works in row bands to
    bound memory.  ``rows`` restricts the output to those canvas rows (returned stacked)."""
    final_w, final_h = final_wh
    ori_h, ori_w, _ = ori_img.shape
    mesh_w, mesh_h = mesh
    off_x, off_y = offset
    rc_all = cell_lookup(final_h, mesh_h)
    cc = cell_lookup(final_w, mesh_w)
    row_ids = np.arange(final_h) if rows is None else np.asarray(rows, dtype=np.int64)
    warped = np.zeros((len(row_ids), final_w, 3), dtype=np.uint8)
    x = (np.arange(final_w) - off_x).astype(np.float64)[None, :]
    for lo in range(0, len(row_ids), band):
        hi = min(len(row_ids), lo + band)
        hc = hinv[rc_all[row_ids[lo:hi]]][:, cc].astype(np.float64)
        y = (row_ids[lo:hi] - off_y).astype(np.float64)[:, None]
        t0 = hc[..., 0, 0] * x + hc[..., 0, 1] * y + hc[..., 0, 2]
        t1 = hc[..., 1, 0] * x + hc[..., 1, 1] * y + hc[..., 1, 2]
        t2 = hc[..., 2, 0] * x + hc[..., 2, 1] * y + hc[..., 2, 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            tx, ty = t0 / t2, t1 / t2
            ok = (0 < tx) & (tx < ori_w) & (0 < ty) & (ty < ori_h)
        ix = np.where(ok, tx, 0).astype(np.int64)
        iy = np.where(ok, ty, 0).astype(np.int64)
        px = ori_img[iy, ix]
        px[~ok] = 0
        warped[lo:hi] = px
    return warped


def stitch(warped, center_img, offset):
    """The commented-out tail of the reference's __main__ (apap.py:259-261): paste the
    centre image on an empty canvas at the offsets, then uniform_blend."""
    off_x, off_y = offset
    ch, cw = center_img.shape[:2]
    dst_temp = np.zeros_like(warped)
    dst_temp[off_y:ch + off_y, off_x:cw + off_x, :] = center_img
    return uniform_blend(warped, dst_temp)


# --------------------------------------------------------------------------------------
# output stage (apap.py:250-265) and parity metrics
# --------------------------------------------------------------------------------------
def invert_normalize_flatten(local_h):
    """Per cell ``H <- inv(H); H /= H[2,2]`` in float32, then transpose the 3x3 and
    flatten to rows of 9 float64 (column-major H^-1).  Reference: apap.py:250-264."""
    h = local_h.copy()
    rows, cols = h.shape[:2]
    for i in range(rows):
        for j in range(cols):
            h[i, j] = np.linalg.inv(h[i, j].copy())
            h[i, j] /= h[i, j, -1, -1]
    return h.transpose(0, 1, 3, 2).astype(np.float64).reshape(-1, 9)


def project(H, pts):
    """Project (K,2) points through (...,3,3) homographies -> (...,K,2), float64."""
    H = np.asarray(H, dtype=np.float64)
    p = np.concatenate([np.asarray(pts, dtype=np.float64), np.ones((len(pts), 1))], axis=1)
    q = np.einsum("...ij,kj->...ki", H, p)
    return q[..., :2] / q[..., 2:3]


def reprojection_rmse_delta(H_a, H_b, pts):
    """Parity metric for per-cell homographies: for every cell, the RMS over ``pts``
    of ``|proj(H_a, p) - proj(H_b, p)|`` in pixels.  Returns the per-cell array."""
    d = project(H_a, pts) - project(H_b, pts)
    return np.sqrt(np.mean(np.sum(d * d, axis=-1), axis=-1))
