"""TEST INFRASTRUCTURE - never imported by the product.

Specification, in numpy, of the float32 estimate + doubt window that the default warp kernel
``k_warp_fast`` decides pixels with (``cvx_proj_amd/csrc/apap_kernels.hip``: ``fast_origin``,
``fast_record``, the row loop of ``k_warp_fast``).  The reference decides a canvas pixel by
``int(tx), int(ty)`` and the strict test ``0 < t < size`` of its float64 coordinates
(``/root/reference/pyviz/apap.py:211-215``); the kernel works on a float32 estimate of those
coordinates and recomputes a pixel exactly when the estimate is within a proven error bound of an
integer.  This file restates the record arithmetic and the kernel's float32 operations so that
``tests/test_warp_fast_bound.py`` can check, on the CPU, the claim the kernel's exactness rests on:

    for every pixel NOT flagged "in doubt":  anchor + (fixed >> 16) == floor(reference coordinate)
    and the reference coordinate is not an integer,

with the hardware's freedom (``v_rcp_f32`` is accurate to 1 ulp, not correctly rounded) played
adversarially.
"""
import numpy as np

FRAC_BITS = 16          # 16.16 fixed point: integer part and fraction are the two 16-bit halves (SDWA operand selects)
UNIT = float(1 << FRAC_BITS)
MAX_SPAN = 254
EPS64 = 2.0 ** -53
EPS32 = 2.0 ** -24


def origin(edges, count):
    """fast_origin for every cell of one axis: (ok, first pixel, span)."""
    e0, e1 = np.asarray(edges[:-1], np.float64), np.asarray(edges[1:], np.float64)
    with np.errstate(invalid="ignore"):
        ok = (e0 > -1.0) & (e1 > e0) & (e0 < 2147483000.0)
        a = np.minimum(np.maximum(np.ceil(e0), 0.0), count)
        b = np.minimum(np.maximum(np.ceil(e1), 0.0), count)
    ok &= b > a
    a = np.where(ok, a, 0.0)
    span = np.where(ok, np.minimum(b - a, MAX_SPAN), 1.0)
    return ok, a.astype(np.int64), span.astype(np.int64)


def record(h, ok, xb, yb, DX, DY):
    """fast_record for arrays of cells: ``h`` (..., 9) float64 inverse, anchor (xb, yb), half-extents DX, DY.
    Returns a dict of float32 / integer arrays; ``thr == 0xffffffff`` marks everything-in-doubt cells."""
    h = np.asarray(h, np.float64)
    h0, h1, h2, h3, h4, h5, h6, h7, h8 = (h[..., k] for k in range(9))
    with np.errstate(all="ignore"):
        t0b = (h1 * yb + h0 * xb) + h2
        t1b = (h4 * yb + h3 * xb) + h5
        t2b = (h7 * yb + h6 * xb) + h8
        S0 = np.abs(h0) * (np.abs(xb) + DX) + np.abs(h1) * (np.abs(yb) + DY) + np.abs(h2)
        S1 = np.abs(h3) * (np.abs(xb) + DX) + np.abs(h4) * (np.abs(yb) + DY) + np.abs(h5)
        S2 = np.abs(h6) * (np.abs(xb) + DX) + np.abs(h7) * (np.abs(yb) + DY) + np.abs(h8)
        at2 = np.abs(t2b)
        g = np.abs(h6) * DX + np.abs(h7) * DY
        good = ok & (at2 > 1e-20) & (at2 < 1e20) & (g <= 0.25 * at2)
        tmin = at2 - g
        rho = (at2 + g) / tmin
        Qx, Qy = t0b / t2b, t1b / t2b
        good &= (np.abs(Qx) < 1073741824.0) & (np.abs(Qy) < 1073741824.0)
        n0x, n0y = np.floor(Qx), np.floor(Qy)
        Ax, Bx = h0 - Qx * h6, h1 - Qx * h7
        Ay, By = h3 - Qy * h6, h4 - Qy * h7
        Mx = (np.abs(Ax) * DX + np.abs(Bx) * DY) / tmin
        My = (np.abs(Ay) * DX + np.abs(By) * DY) / tmin
        Smax = np.maximum(Mx, My) + 1.25 * rho
        good &= Smax < 500.0
        E32 = (6.5 + 3.0 * rho) * EPS32 * Smax
        E64 = 16.0 * EPS64 * (np.maximum(S0, S1) + (np.maximum(np.abs(Qx), np.abs(Qy)) + Smax) * (S2 + at2)) / tmin
        # the estimate is shifted up by dE, strictly more than its error bound; the conversion is a FLOOR (v_cvt_flr_i32_f32), so
        # with S = the float32 value, F = floor(S), m = F >> 16, frac = F & 0xffff and v = the reference coordinate - n0:
        #     v in (S u - 2 dE, S u),  S in [F, F + 1)   =>   m < v < m + 1  as soon as  frac u >= 2 dE   (u = 2^-16)
        dE = (E32 + E64) * (1.0 + 2.0 ** -20) + 2.0 ** -40
        thr16 = np.maximum(np.ceil(2.0 * dE * UNIT), 1.0)
        good &= dE < 0.125
        good &= np.isfinite(dE)
        du = 0.5 * thr16              # (half the window in units of 2^-16: what the reports print)
        fx, fy = (Qx - n0x) + dE, (Qy - n0y) + dE
    z = np.zeros_like(t2b)
    f32 = lambda v: np.where(good, v, z).astype(np.float32)   # noqa: E731
    return {
        "ax": f32(UNIT * (fx * h6 + Ax)), "bx": f32(UNIT * (fx * h7 + Bx)), "cx": f32(UNIT * (fx * t2b)),
        "ay": f32(UNIT * (fy * h6 + Ay)), "by": f32(UNIT * (fy * h7 + By)), "cy": f32(UNIT * (fy * t2b)),
        "t2b": f32(t2b), "h6": f32(h6), "h7": f32(h7),
        "n0x": np.where(good, n0x, 0).astype(np.int64), "n0y": np.where(good, n0y, 0).astype(np.int64),
        "thr": np.where(good, np.where(good, thr16, 0).astype(np.int64), 0xffffffff).astype(np.int64),
        "good": good, "E": np.where(good, E32 + E64, np.inf), "du": np.where(good, du, np.inf),
    }


def _fma32(a, b, c):
    """One float32 FMA: the exact a*b + c (float64 holds the product of two float32 exactly; the sum is
    rounded twice, which moves the result by at most one part in 2^29 of a float32 ulp)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def estimate(rec, dx, dy, rcp_ulps=0):
    """The kernel's float32 arithmetic for pixels at (dx, dy) from their cell's anchor: returns
    (ix, iy, doubt).  ``rcp_ulps`` (array or int, -1/0/+1) perturbs the reciprocal by whole float32 ulps:
    v_rcp_f32 is accurate to 1 ulp, the correctly rounded value is only one of its legal results."""
    dxf, dyf = np.asarray(dx, np.float32), np.asarray(dy, np.float32)
    nx = _fma32(rec["bx"], dyf, _fma32(rec["ax"], dxf, rec["cx"]))
    ny = _fma32(rec["by"], dyf, _fma32(rec["ay"], dxf, rec["cy"]))
    den = _fma32(rec["h7"], dyf, _fma32(rec["h6"], dxf, rec["t2b"]))
    with np.errstate(all="ignore"):
        rc = (np.float32(1.0) / den).astype(np.float32)
        rc = np.where(np.isfinite(rc), rc + np.spacing(np.abs(rc)) * np.asarray(rcp_ulps, np.float32), rc).astype(np.float32)
        sx, sy = (nx * rc).astype(np.float32), (ny * rc).astype(np.float32)

        def to_int(v):     # v_cvt_flr_i32_f32: floor, saturating, NaN -> 0
            v = np.where(np.isnan(v), np.float32(0), v)
            return np.clip(np.floor(v.astype(np.float64)), -2147483648.0, 2147483647.0).astype(np.int64)
        fx, fy = to_int(sx), to_int(sy)
    lo = np.minimum(fx & 0xffff, fy & 0xffff)        # the fractions: the low halves (two's complement: floor semantics)
    doubt = lo < rec["thr"]
    return rec["n0x"] + (fx >> FRAC_BITS), rec["n0y"] + (fy >> FRAC_BITS), doubt


def reference_coords(h, x, y):
    """apap.py:172-184 on float64 (the float32 inverse widened): (H @ [x, y, 1]) / third component."""
    h = np.asarray(h, np.float64)
    with np.errstate(all="ignore"):
        t0 = (h[..., 0] * x + h[..., 1] * y) + h[..., 2]
        t1 = (h[..., 3] * x + h[..., 4] * y) + h[..., 5]
        t2 = (h[..., 6] * x + h[..., 7] * y) + h[..., 8]
        return t0 / t2, t1 / t2
