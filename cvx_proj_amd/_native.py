"""ctypes binding of ``libapap_hip.so`` (the C ABI declared in ``include/apap_hip.h``).

This is the whole Python<->native boundary: plain pointers and sizes, no torch types.
The library is built in-tree by ``cvx_proj_amd/csrc/Makefile`` (see
``__graft_entry__.build``).  There is no CPU fallback: if the library is missing or no
gfx950 device is visible, compute calls raise :class:`ApapError`.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# APAP_HIP_LIB: load another build of the library (A/B tooling: tools/ab_build.sh); default in-tree
LIB_PATH = os.environ.get("APAP_HIP_LIB") or os.path.join(_HERE, "libapap_hip.so")

OK, ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_SINGULAR, ERR_INDEX, ERR_WORKSPACE = range(7)
ABI_VERSION = 6          # APAP_ABI_VERSION of include/apap_hip.h
# kernel slots of apap_ctx_profile_read (include/apap_hip.h)
PROF_NAMES = ("assemble", "eigen", "invert", "lut", "warp", "eq_hist", "eq_apply", "ransac")
PROF_SLOTS = len(PROF_NAMES)
TABLE_STRIDE = 32
DENORM_DOUBLES = 36
VARIANT_AUTO, VARIANT_VALU, VARIANT_MFMA, VARIANT_MFMA4, VARIANT_MFMA4X2 = 0, 1, 2, 3, 4
EIGEN_AUTO, EIGEN_JACOBI, EIGEN_INVERSE_ITERATION = 0, 1, 2
# options of a context (include/apap_hip.h)
OPT_SOLVER_VARIANT, OPT_EIGEN_SOLVER, OPT_CAREFUL, OPT_PROFILE, OPT_WANT_WAVES, OPT_WARP_ROWS, OPT_WEIGHT_CHUNK_KB, \
    OPT_FUSED_MAX_CELLS, OPT_WARP_FAST, OPT_OVERLAP_PCIE, OPT_PLAN_CELLS, OPT_MOMENTS, OPT_WEIGHTS_F32 = range(13)


class ApapError(RuntimeError):
    """A native call failed; ``code`` is one of the APAP_ERR_* values.  The conditions the
    reference signals with a specific Python exception raise a subclass that is ALSO that
    exception, so ``except np.linalg.LinAlgError`` / ``except IndexError`` / ``except ValueError``
    around the reference's class keep working around this one."""

    def __init__(self, code, message):
        super().__init__(f"[apap_hip error {code}] {message}")
        self.code = code


class ApapSingularError(ApapError, np.linalg.LinAlgError):
    """APAP_ERR_SINGULAR: ``numpy.linalg.inv`` of a cell raised LinAlgError("Singular matrix")
    (apap.py:165-166,203,252)."""


class ApapIndexError(ApapError, IndexError):
    """APAP_ERR_INDEX: ``np.where(i < mesh_h)[0][0]`` found no edge above a canvas index
    (apap.py:207,209)."""


class ApapValueError(ApapError, ValueError):
    """APAP_ERR_INVALID_ARG: what the reference's shape unpacking rejects with ValueError
    (apap.py:129-130,197,199)."""


_ERROR_CLASSES = {ERR_SINGULAR: ApapSingularError, ERR_INDEX: ApapIndexError, ERR_INVALID_ARG: ApapValueError}


_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int)
_vp = C.c_void_p

# name -> (restype, argtypes).  Must list every symbol include/apap_hip.h declares;
# tests/test_capi_symbols.py parses the header and checks this table against it.
SIGNATURES = {
    "apap_last_error": (C.c_char_p, []),
    "apap_version": (C.c_char_p, []),
    "apap_abi_version": (C.c_int, []),
    "apap_device_count": (C.c_int, []),
    "apap_ctx_create": (C.c_void_p, []),
    "apap_ctx_destroy": (None, [_vp]),
    "apap_ctx_set_option": (C.c_int, [_vp, C.c_int, C.c_int]),
    "apap_ctx_get_option": (C.c_int, [_vp, C.c_int, _i32p]),
    "apap_ctx_profile_read": (C.c_int, [_vp, _f32p, _i32p]),
    "apap_host_prepare": (C.c_int, [_f32p, _f32p, C.c_int] + [_f32p] * 10),
    "apap_host_dlt_rows": (C.c_int, [_f32p, _f32p, C.c_int, _f32p]),
    "apap_host_prepare_pts": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int] + [_f32p] * 6 + [_f64p] * 4),
    "apap_host_dlt_rows_pts": (C.c_int, [_f64p, _f64p, C.c_int, C.c_int, _f32p]),
    "apap_host_build_table_rows": (C.c_int, [_f64p, _f32p, C.c_int, _f64p]),
    "apap_host_build_table": (C.c_int, [_f32p, _f32p, _f32p, C.c_int, _f64p]),
    "apap_host_build_table24": (C.c_int, [_f64p, _f32p, C.c_int, _f64p]),
    "apap_host_build_denorm": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f64p]),
    "apap_local_homography": (C.c_int, [_vp, _f32p, _f32p, C.c_int, _f64p, C.c_int, C.c_int, C.c_double,
                                        C.c_double, _f32p, _f64p, C.c_int]),
    "apap_local_weights": (C.c_int, [_vp, _f32p, C.c_int, _f64p, C.c_int, C.c_double, C.c_double, _f64p, C.c_int]),
    "apap_local_homography_pts": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int, _f64p, C.c_int, C.c_int, C.c_double,
                                            C.c_double, _f32p, _f64p, C.c_int]),
    "apap_local_weights_pts": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _f64p, C.c_int, C.c_double, C.c_double, _f64p, C.c_int]),
    "apap_local_warp": (C.c_int, [_vp, _u8p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, _f64p, C.c_int, _f64p,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, _f32p, C.c_int]),
    "apap_local_warp_f64": (C.c_int, [_vp, _u8p, C.c_int, C.c_int, _f64p, C.c_int, C.c_int, _f64p, C.c_int, _f64p,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, _f64p, C.c_int]),
    "apap_local_stitch": (C.c_int, [_vp, _u8p, C.c_int, C.c_int, _u8p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, _f64p,
                                    C.c_int, _f64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, _f32p, C.c_int]),
    "apap_warp_coords": (C.c_int, [_vp, _f32p, C.c_int, C.c_int, _f64p, C.c_int, _f64p, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, _f64p, C.c_int]),
    "apap_invert_normalize_flatten": (C.c_int, [_vp, _f32p, C.c_int, _f64p, C.c_int]),
    "apap_uniform_blend": (C.c_int, [_vp, _u8p, _u8p, C.c_int, C.c_int, _u8p, C.c_int]),
    "apap_solve_workspace_bytes": (C.c_size_t, [_vp, C.c_int, C.c_int]),
    "apap_solve_device": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, C.c_double, C.c_double, _vp, _vp, _vp,
                                    C.c_size_t, _vp]),
    "apap_solve_batch_workspace_bytes": (C.c_size_t, [_vp, C.c_int, C.c_int, C.c_int]),
    "apap_solve_batch_device": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_longlong, C.c_int, C.c_double, C.c_double, _vp, _vp,
                                          C.c_int, _vp, C.c_size_t, _vp]),
    "apap_solve_warp_batch_device": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_longlong, C.c_double, C.c_double, _vp, _vp, C.c_int, _vp,
                                               C.c_size_t, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.c_int, _vp, C.c_size_t, _vp, _vp]),
    "apap_weights_device": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, C.c_double, C.c_double, _vp, _vp]),
    "apap_warp_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "apap_warp_device": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_size_t, _vp, _vp]),
    "apap_warp_f64_device": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_size_t, _vp, _vp]),
    "apap_warp_rows_device": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_size_t, _vp,
                                        _vp]),
    "apap_warp_batch_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "apap_warp_batch_device": (C.c_int, [_vp, _vp, C.c_longlong, C.c_int, C.c_int, _vp, C.c_longlong, C.c_int, C.c_int, _vp, C.c_int,
                                         C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         _vp, C.c_longlong, _vp, C.c_int, C.c_int, _vp, C.c_size_t, _vp, _vp]),
    "apap_stitch_device": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_int,
                                     _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_size_t, _vp, _vp]),
    "apap_warp_coords_device": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, _vp, _vp, C.c_size_t, _vp, _vp]),
    "apap_flatten_device": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp]),
    "apap_blend_device": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp]),
    "apap_equalize_hist": (C.c_int, [_vp, _u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int]),
    "apap_equalize_workspace_bytes": (C.c_size_t, [C.c_int]),
    "apap_equalize_hist_device": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_size_t, _vp]),
    "apap_find_homography_ransac": (C.c_int, [_vp, _f32p, _f32p, C.c_int, C.c_double, C.c_int, C.c_ulonglong, _f64p, _u8p,
                                              _i32p, C.c_int]),
    "apap_ransac_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "apap_ransac_device": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_double, C.c_int, C.c_ulonglong, _vp, _vp, _vp, _vp,
                                     C.c_size_t, _vp]),
}

_lib = None


def lib():
    """Load the shared library once.  Raises ApapError (never falls back) if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ApapError(ERR_NO_DEVICE, f"{LIB_PATH} not built; run `python -c 'import __graft_entry__ as g; "
                                           "g.build()'` or `make -C cvx_proj_amd/csrc`")
        # One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64.so (same SONAME as /opt/rocm's, different
        # file).  When torch is ALREADY imported its copy is in the process and this library binds to it; otherwise the RPATH
        # to /opt/rocm/lib is used and NO torch is imported on the library's behalf (2.3 s against 0.4 s of start-up for a
        # command that needs none: python -m cvx_proj_amd.apap).  The rule for a process that uses both: import torch BEFORE
        # the first call into this module (cvx_proj_amd.pipeline and cvx_proj_amd.dist do so at their top; a later
        # `import torch` would find /opt/rocm's runtime under its own SONAME and see no GPU).  APAP_HIP_PRELOAD_TORCH=1
        # restores the old behaviour (torch imported here first, if it is installed).
        if "torch" not in sys.modules and os.environ.get("APAP_HIP_PRELOAD_TORCH", "0") == "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        handle = C.CDLL(LIB_PATH)
        # a build of another ABI generation keeps the symbol names but not the argument lists: refuse it
        got = handle.apap_abi_version() if hasattr(handle, "apap_abi_version") else 1
        if got != ABI_VERSION:
            raise ApapError(ERR_INVALID_ARG, f"{LIB_PATH} is ABI generation {got}, this binding needs {ABI_VERSION}: rebuild "
                                             "it (make -C cvx_proj_amd/csrc)")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(code):
    if code != OK:
        raise _ERROR_CLASSES.get(code, ApapError)(code, lib().apap_last_error().decode("utf-8", "replace"))


class Context:
    """``apap_ctx`` of include/apap_hip.h: solver options, per-kernel profiling and the pool of
    device buffers that host-buffer calls reuse.  The library has no process-wide mutable state;
    pass a context as ``ctx=`` to any wrapper below (``None`` = built-in defaults).  Not to be
    shared between threads; use one per thread."""

    def __init__(self, **options):
        self._h = lib().apap_ctx_create()
        if not self._h:
            raise ApapError(ERR_HIP, "apap_ctx_create failed")
        for k, v in options.items():
            self.set(k, v)

    _NAMES = {"variant": OPT_SOLVER_VARIANT, "eigen": OPT_EIGEN_SOLVER, "careful": OPT_CAREFUL, "profile": OPT_PROFILE,
              "want_waves": OPT_WANT_WAVES, "warp_rows": OPT_WARP_ROWS, "weight_chunk_kb": OPT_WEIGHT_CHUNK_KB,
              "fused_max_cells": OPT_FUSED_MAX_CELLS, "warp_fast": OPT_WARP_FAST, "overlap_pcie": OPT_OVERLAP_PCIE,
              "plan_cells": OPT_PLAN_CELLS, "moments": OPT_MOMENTS, "weights_f32": OPT_WEIGHTS_F32}

    def set(self, name, value):
        check(lib().apap_ctx_set_option(self._h, self._NAMES[name], int(value)))
        return self

    def get(self, name):
        v = C.c_int(0)
        check(lib().apap_ctx_get_option(self._h, self._NAMES[name], C.byref(v)))
        return v.value

    def profile_read(self):
        """``{kernel slot name: (milliseconds, launches)}`` since the previous read."""
        ms = (C.c_float * PROF_SLOTS)()
        cnt = (C.c_int * PROF_SLOTS)()
        check(lib().apap_ctx_profile_read(self._h, ms, cnt))
        return {k: (ms[i], cnt[i]) for i, k in enumerate(PROF_NAMES)}

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            lib().apap_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001  (interpreter shutdown)
            pass


def _h(ctx):
    """ctypes handle of ``ctx`` (a Context, a raw handle, or None)."""
    if ctx is None:
        return None
    return C.c_void_p(ctx.handle if isinstance(ctx, Context) else ctx)


def last_error():
    return lib().apap_last_error().decode("utf-8", "replace")


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype)) if a is not None else None


def as_f32(a, shape_tail=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape_tail is not None and tuple(a.shape[-len(shape_tail):]) != tuple(shape_tail):
        raise ValueError(f"expected trailing shape {shape_tail}, got {a.shape}")
    return a


def as_points(a):
    """Keypoints in the dtype the reference would compute with: float32 stays float32 (what utils.get_features returns);
    float64 and integer arrays are taken as float64 (numpy's own reductions promote integers to float64; nothing in
    apap.py:35-100 casts its argument) - both pinned bit for bit by tests/golden/f64pts_ref.npz.  Any OTHER dtype (float16,
    longdouble, bool, ...) is widened to float64 as well, which is NOT what numpy does with it in the reference (a float16
    set mixes with the float32 padding to float32 there): accepted, but without the bit-compatibility claim.
    Returns (contiguous (n, 2) array, is_float64)."""
    a = np.asarray(a)
    a = np.ascontiguousarray(a, dtype=np.float32 if a.dtype == np.float32 else np.float64)
    if a.shape[-1:] != (2,):
        raise ValueError(f"expected trailing shape (2,), got {a.shape}")
    return a, int(a.dtype == np.float64)


def _vptr(a):
    return a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------ host-only helpers
def host_prepare(src, dst):
    """C restatement of the set-up in apap.py:132-140,165-166, in the dtype of each keypoint set.  Returns a dict: N1 N2 C1 C2
    iC2 iN2 (3x3 float32) and nf1 nf2 cf1 cf2 (n x 2; float32 for a float32 set, float64 for a float64 one)."""
    src, s64 = as_points(src)
    dst, d64 = as_points(dst)
    if src.shape != dst.shape or src.ndim != 2:
        raise ValueError(f"src/dst must both be (n, 2); got {src.shape} and {dst.shape}")
    n = src.shape[0]
    out = {k: np.empty((3, 3), np.float32) for k in ("N1", "N2", "C1", "C2", "iC2", "iN2")}
    if not s64 and not d64:
        out.update({k: np.empty((n, 2), np.float32) for k in ("nf1", "nf2", "cf1", "cf2")})
        check(lib().apap_host_prepare(_ptr(src, C.c_float), _ptr(dst, C.c_float), n,
                                      *[_ptr(out[k], C.c_float) for k in
                                        ("N1", "N2", "C1", "C2", "iC2", "iN2", "nf1", "nf2", "cf1", "cf2")]))
        return out
    wide = {k: np.empty((n, 2), np.float64) for k in ("nf1", "nf2", "cf1", "cf2")}
    check(lib().apap_host_prepare_pts(_vptr(src), s64, _vptr(dst), d64, n,
                                      *[_ptr(out[k], C.c_float) for k in ("N1", "N2", "C1", "C2", "iC2", "iN2")],
                                      *[_ptr(wide[k], C.c_double) for k in ("nf1", "nf2", "cf1", "cf2")]))
    for k, is64 in (("nf1", s64), ("cf1", s64), ("nf2", d64), ("cf2", d64)):
        out[k] = wide[k] if is64 else wide[k].astype(np.float32)        # exact: the values ARE float32
    return out


def host_dlt_rows(cf1, cf2):
    """APAP.matrix_generate (apap.py:103-119): float32 rows; a float64 operand makes the products float64 products rounded
    once on the store, two float32 operands float32 products."""
    cf1, a64 = as_points(cf1)
    cf2, b64 = as_points(cf2)
    n = cf1.shape[0]
    aa = np.empty((2 * n, 9), np.float32)
    if not a64 and not b64:
        check(lib().apap_host_dlt_rows(_ptr(cf1, C.c_float), _ptr(cf2, C.c_float), n, _ptr(aa, C.c_float)))
    else:
        cf1, cf2 = np.ascontiguousarray(cf1, np.float64), np.ascontiguousarray(cf2, np.float64)
        check(lib().apap_host_dlt_rows_pts(_ptr(cf1, C.c_double), _ptr(cf2, C.c_double), n, 1, _ptr(aa, C.c_float)))
    return aa


def _out_f64(out, shape):
    """``out`` as the function's result buffer (e.g. page-locked staging memory of the caller's), or a fresh array."""
    if out is None:
        return np.empty(shape, np.float64)
    if not (isinstance(out, np.ndarray) and out.dtype == np.float64 and out.shape == tuple(shape) and out.flags.c_contiguous):
        raise ValueError(f"out must be a contiguous float64 array of shape {tuple(shape)}")
    return out


def host_build_table(src, cf1, cf2, out=None, moments=30):
    """The device keypoint table of ``apap_solve_device`` and its batch forms.  ``moments`` = 30 (default): the 30 distinct
    entries of ``r1 r1^T + r2 r2^T`` of the reference's float32 DLT rows; 24: the exact-product table of a context with
    ``moments=24`` (``apap_host_build_table24``; pass ``ctx.get("moments")``)."""
    src, s64 = as_points(src)
    cf1, a64 = as_points(cf1)
    cf2, b64 = as_points(cf2)
    n = src.shape[0]
    table = _out_f64(out, (n, TABLE_STRIDE))
    if moments not in (24, 30):
        raise ValueError(f"moments must be 30 or 24, got {moments}")
    if moments == 30 and not (s64 or a64 or b64):
        check(lib().apap_host_build_table(_ptr(src, C.c_float), _ptr(cf1, C.c_float), _ptr(cf2, C.c_float), n,
                                          _ptr(table, C.c_double)))
    else:       # from the DLT rows themselves and the source keypoints as float64
        aa = host_dlt_rows(cf1, cf2)
        src = np.ascontiguousarray(src, np.float64)
        build = lib().apap_host_build_table24 if moments == 24 else lib().apap_host_build_table_rows
        check(build(_ptr(src, C.c_double), _ptr(aa, C.c_float), n, _ptr(table, C.c_double)))
    return table


def host_build_denorm(iC2, C1, iN2, N1, out=None):
    mats = [as_f32(m, (3, 3)) for m in (iC2, C1, iN2, N1)]
    out = _out_f64(out, (DENORM_DOUBLES,))
    check(lib().apap_host_build_denorm(*[_ptr(m, C.c_float) for m in mats], _ptr(out, C.c_double)))
    return out


# ---------------------------------------------------------------- host-buffer compute
def local_homography(src, dst, vertices, gamma, sigma, want_weights=True, device=-1, ctx=None):
    src, s64 = as_points(src)
    dst, d64 = as_points(dst)
    if src.ndim != 2 or src.shape != dst.shape:
        raise ValueError(f"src/dst must both be (n, 2); got {src.shape} and {dst.shape}")
    vertices = np.ascontiguousarray(vertices, dtype=np.float64)
    rows, cols, two = vertices.shape      # ValueError on a wrong rank, like apap.py:130
    if two != 2:
        raise ValueError(f"vertices must be (rows, cols, 2); got {vertices.shape}")
    n = src.shape[0]
    H = np.empty((rows, cols, 3, 3), np.float32)
    W = np.empty((rows, cols, n), np.float64) if want_weights else None
    check(lib().apap_local_homography_pts(_h(ctx), _vptr(src), s64, _vptr(dst), d64, n, _ptr(vertices, C.c_double),
                                          rows, cols, float(gamma), float(sigma), _ptr(H, C.c_float),
                                          _ptr(W, C.c_double), device))
    return H, W


def local_weights(src, points, gamma, sigma, device=-1, ctx=None):
    """``max(exp(-|p - s| / sigma^2), gamma)`` for every (sample point p, keypoint s): ``points`` (..., 2) float64
    -> (..., n) float64.  The weights of reference apap.py:150-153 for any subset of the mesh."""
    src, s64 = as_points(src)
    pts = np.ascontiguousarray(points, dtype=np.float64)
    if pts.shape[-1:] != (2,):
        raise ValueError(f"points must be (..., 2); got {pts.shape}")
    n, cells = src.shape[0], pts.size // 2
    W = np.empty(pts.shape[:-1] + (n,), np.float64)
    if cells:
        check(lib().apap_local_weights_pts(_h(ctx), _vptr(src), s64, n, _ptr(pts, C.c_double), cells, float(gamma),
                                           float(sigma), _ptr(W, C.c_double), device))
    return W


def local_warp(img, H, mesh_w, mesh_h, final_w, final_h, off_x, off_y, want_inverse=True, device=-1, ctx=None, out=None):
    """``out``: the caller's own (final_h, final_w, 3) uint8 canvas (e.g. page-locked memory, which the library then neither
    registers nor stages) instead of a fresh array."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    img_h, img_w, ch = img.shape
    if ch != 3:
        raise ValueError(f"image must be (h, w, 3); got {img.shape}")
    mesh_w = np.ascontiguousarray(mesh_w, dtype=np.float64)
    mesh_h = np.ascontiguousarray(mesh_h, dtype=np.float64)
    if out is None:
        out = np.empty((final_h, final_w, 3), np.uint8)
    elif not (isinstance(out, np.ndarray) and out.shape == (final_h, final_w, 3) and out.dtype == np.uint8 and out.flags.c_contiguous):
        raise ValueError(f"out must be a contiguous uint8 array of shape {(final_h, final_w, 3)}")
    if isinstance(H, np.ndarray) and H.dtype == np.float64:
        # the reference inverts and multiplies in the grid's own dtype (apap.py:201-203,210-213): a
        # float64 grid is not rounded to float32 on the way
        H = np.ascontiguousarray(H)
        if H.shape[-2:] != (3, 3):
            raise ValueError(f"expected trailing shape (3, 3), got {H.shape}")
        rows, cols = H.shape[:2]
        Hinv = np.empty_like(H) if want_inverse else None
        check(lib().apap_local_warp_f64(_h(ctx), _ptr(img, C.c_uint8), img_h, img_w, _ptr(H, C.c_double), rows, cols,
                                        _ptr(mesh_w, C.c_double), mesh_w.size, _ptr(mesh_h, C.c_double), mesh_h.size,
                                        int(final_w), int(final_h), int(off_x), int(off_y), _ptr(out, C.c_uint8),
                                        _ptr(Hinv, C.c_double), device))
        return out, Hinv
    H = as_f32(H, (3, 3))
    rows, cols = H.shape[:2]
    Hinv = np.empty_like(H) if want_inverse else None
    check(lib().apap_local_warp(_h(ctx), _ptr(img, C.c_uint8), img_h, img_w, _ptr(H, C.c_float), rows, cols,
                                _ptr(mesh_w, C.c_double), mesh_w.size, _ptr(mesh_h, C.c_double), mesh_h.size,
                                int(final_w), int(final_h), int(off_x), int(off_y), _ptr(out, C.c_uint8),
                                _ptr(Hinv, C.c_float), device))
    return out, Hinv


def local_stitch(img, center, H, mesh_w, mesh_h, final_w, final_h, off_x, off_y, want_inverse=False, device=-1, ctx=None):
    """Fused local_warp + paste of ``center`` at the offsets + uniform_blend."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    center = np.ascontiguousarray(center, dtype=np.uint8)
    if img.ndim != 3 or img.shape[2] != 3 or center.ndim != 3 or center.shape[2] != 3:
        raise ValueError(f"images must be (h, w, 3); got {img.shape} and {center.shape}")
    H = as_f32(H, (3, 3))
    rows, cols = H.shape[:2]
    mesh_w = np.ascontiguousarray(mesh_w, dtype=np.float64)
    mesh_h = np.ascontiguousarray(mesh_h, dtype=np.float64)
    out = np.empty((final_h, final_w, 3), np.uint8)
    Hinv = np.empty_like(H) if want_inverse else None
    check(lib().apap_local_stitch(_h(ctx), _ptr(img, C.c_uint8), img.shape[0], img.shape[1], _ptr(center, C.c_uint8),
                                  center.shape[0], center.shape[1], _ptr(H, C.c_float), rows, cols,
                                  _ptr(mesh_w, C.c_double), mesh_w.size, _ptr(mesh_h, C.c_double), mesh_h.size,
                                  int(final_w), int(final_h), int(off_x), int(off_y), _ptr(out, C.c_uint8),
                                  _ptr(Hinv, C.c_float), device))
    return out, Hinv


def warp_coords(H, mesh_w, mesh_h, final_w, final_h, off_x, off_y, device=-1, ctx=None):
    H = as_f32(H, (3, 3))
    rows, cols = H.shape[:2]
    mesh_w = np.ascontiguousarray(mesh_w, dtype=np.float64)
    mesh_h = np.ascontiguousarray(mesh_h, dtype=np.float64)
    coords = np.empty((final_h, final_w, 2), np.float64)
    check(lib().apap_warp_coords(_h(ctx), _ptr(H, C.c_float), rows, cols, _ptr(mesh_w, C.c_double), mesh_w.size,
                                 _ptr(mesh_h, C.c_double), mesh_h.size, int(final_w), int(final_h), int(off_x),
                                 int(off_y), _ptr(coords, C.c_double), device))
    return coords


def invert_normalize_flatten(H, device=-1, ctx=None):
    H = as_f32(H, (3, 3))
    cells = H.size // 9
    out = np.empty((cells, 9), np.float64)
    check(lib().apap_invert_normalize_flatten(_h(ctx), _ptr(H, C.c_float), cells, _ptr(out, C.c_double), device))
    return out


def uniform_blend(img1, img2, device=-1, ctx=None):
    a = np.ascontiguousarray(img1, dtype=np.uint8)
    b = np.ascontiguousarray(img2, dtype=np.uint8)
    if a.shape != b.shape or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError(f"images must share shape (h, w, 3); got {a.shape} and {b.shape}")
    out = np.empty_like(a)
    check(lib().apap_uniform_blend(_h(ctx), _ptr(a, C.c_uint8), _ptr(b, C.c_uint8), a.shape[0], a.shape[1],
                                   _ptr(out, C.c_uint8), device))
    return out


def equalize_hist(img, device=-1, ctx=None):
    """Per-channel ``cv.equalizeHist`` of an (h, w) or (h, w, c) uint8 image, c <= 4."""
    a = np.ascontiguousarray(img, dtype=np.uint8)
    if a.ndim not in (2, 3) or (a.ndim == 3 and not 1 <= a.shape[2] <= 4) or a.size == 0:
        raise ValueError(f"image must be (h, w) or (h, w, 1..4) uint8 and non-empty; got {a.shape}")
    channels = 1 if a.ndim == 2 else a.shape[2]
    out = np.empty_like(a)
    check(lib().apap_equalize_hist(_h(ctx), _ptr(a, C.c_uint8), a.shape[0], a.shape[1], channels, _ptr(out, C.c_uint8), device))
    return out


WARP_GEOMETRY, WARP_CELLS, WARP_GATHER, WARP_ALL = 1, 2, 4, 7      # phases of apap_warp_batch_device
RANSAC_ITERATIONS = 2048                 # include/apap_hip.h
RANSAC_SEED = 0x5EEDC0DE5EEDC0DE


def find_homography_ransac(src, dst, thresh=5.0, iterations=RANSAC_ITERATIONS, seed=RANSAC_SEED, device=-1, ctx=None):
    """``cv.findHomography(src, dst, cv.RANSAC, thresh)``: ``(H (3, 3) float64 or None, mask (n, 1) uint8)``."""
    s = np.ascontiguousarray(src, dtype=np.float32).reshape(-1, 2)
    d = np.ascontiguousarray(dst, dtype=np.float32).reshape(-1, 2)
    if s.shape != d.shape:
        raise ValueError(f"src and dst must have the same number of points; got {s.shape} and {d.shape}")
    H = np.zeros(9, dtype=np.float64)
    mask = np.zeros(len(s), dtype=np.uint8)
    inliers = C.c_int(0)
    check(lib().apap_find_homography_ransac(_h(ctx), _ptr(s, C.c_float), _ptr(d, C.c_float), len(s), float(thresh), int(iterations),
                                            C.c_ulonglong(seed), _ptr(H, C.c_double), _ptr(mask, C.c_uint8),
                                            C.byref(inliers), device))
    if inliers.value < 4:
        return None, mask.reshape(-1, 1)
    return H.reshape(3, 3), mask.reshape(-1, 1)
