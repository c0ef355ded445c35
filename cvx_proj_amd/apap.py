"""Drop-in for the reference's ``pyviz/apap.py``: same class, same methods, same CLI
and the same ``.mat`` output, with the two hot loops running on an MI355X.

``APAP.local_homography`` (reference apap.py:121-169) and ``APAP.local_warp``
(apap.py:186-217) call ``libapap_hip.so`` through :mod:`cvx_proj_amd._native`; the small
static helpers keep their numpy form because they are part of the public surface and
run once per pair.  There is no CPU fallback for the two loops.

Command line (reference apap.py:220-265 takes ``[case_idx] [img_idx]`` and reads a
dataset that is not distributed; the positional arguments and the output file are kept,
the inputs come from ``--pair``)::

    python -m cvx_proj_amd.apap [case_idx] [img_idx] [--cases 1-4 --imgs 1,2,4,5]
                                [--data-root DIR | --pair pair.npz | --synth C1]
                                [--config configs/case1.txt] [--out-prefix ../diff_1/results/]
                                [--mesh-size 100] [--gamma 0.5] [--sigma 100] [--warp out.npy | --stitch out.npy]
                                [--resident] [--timing]

The command goes through the host-buffer entry points of the C ABI (``apap_local_homography_pts``, ``apap_local_warp``,
``apap_invert_normalize_flatten``): no torch in the process - a pair is ~1 ms of GPU work, importing torch costs 2 s.
``--resident`` takes :mod:`cvx_proj_amd.pipeline` instead (torch for device memory; what a caller with many pairs in flight
uses).
"""
from __future__ import annotations

import os
import sys

import numpy as np

from . import _native
from .apap_utils import final_size, get_mesh, get_vertice, uniform_blend  # noqa: F401  (re-exported like the reference)

__all__ = ["APAP", "LazyWeights", "get_mesh", "get_vertice", "final_size", "uniform_blend", "save2mat", "run_pair",
           "run_pair_by_calls", "main"]


class LazyWeights:
    """The second return value of ``APAP.local_homography`` (reference apap.py:144,153,169: ``local_weight``,
    float64 ``(rows, cols, n)``), computed on demand.

    The tensor is 8 n bytes per cell - 640 MB for a 200 x 200 mesh and 2000 keypoints, 6.4 GB at 400 x 400 x
    5000 - and the reference's own caller never looks at it (apap.py:242).  This object has its ``shape``,
    ``dtype``, ``ndim``, ``size`` and ``nbytes``; indexing the mesh axes (``W[i, j]``, ``W[2:4]``,
    ``W[:, 7, :50]``) computes exactly the cells asked for on the GPU; ``np.asarray(W)``, any numpy function,
    arithmetic and every ndarray method materialise the whole tensor once and keep it.  The values are those of
    an eager call (the same kernel): ``max(exp(-|vertex - keypoint| / sigma^2), gamma)``."""

    dtype = np.dtype(np.float64)

    def __init__(self, src, vertices, gamma, sigma, device=-1, ctx=None):
        self._src = _native.as_points(src)[0].copy()      # float32 stays float32, anything else float64 (apap.py:150)
        self._vertices = np.array(vertices, dtype=np.float64)       # a copy: the caller may reuse its buffer
        self._par = (float(gamma), float(sigma), device, ctx)
        self._full = None
        self.shape = tuple(self._vertices.shape[:2]) + (self._src.shape[0],)

    ndim = 3
    size = property(lambda self: int(np.prod(self.shape)))
    nbytes = property(lambda self: self.size * 8)

    def _compute(self, points):
        gamma, sigma, device, ctx = self._par
        return _native.local_weights(self._src, points, gamma, sigma, device=device, ctx=ctx)

    def materialize(self):
        """The whole ``(rows, cols, n)`` float64 array (computed once)."""
        if self._full is None:
            self._full = self._compute(self._vertices)
        return self._full

    def __array__(self, dtype=None, copy=None):
        a = self.materialize()
        if dtype is not None and np.dtype(dtype) != a.dtype:
            return a.astype(dtype)
        return a.copy() if copy else a

    def __getitem__(self, idx):
        if self._full is not None:
            return self._full[idx]
        key = idx if isinstance(idx, tuple) else (idx,)
        simple = lambda k: isinstance(k, (int, np.integer, slice))      # noqa: E731
        if len(key) <= 3 and all(simple(k) for k in key):
            # pick the cells with the first two indices, compute only those, then apply the keypoint index
            cells = self._vertices[key[:2]]
            w = self._compute(cells)
            return w[(Ellipsis,) + key[2:]] if len(key) == 3 else w
        return self.materialize()[idx]

    def __setitem__(self, idx, value):
        """A caller that writes into the reference's array (``W[i, j] = ...``) gets the real array from then on."""
        self.materialize()[idx] = value

    def __len__(self):
        return self.shape[0]

    def __iter__(self):
        return (self[i] for i in range(self.shape[0]))

    def __repr__(self):
        state = "materialised" if self._full is not None else "lazy"
        return f"LazyWeights(shape={self.shape}, dtype=float64, {state})"

    def __getattr__(self, name):        # ndarray methods and attributes: sum, mean, T, reshape, astype, ...
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)


def _delegate(op):
    def method(self, *args):
        return getattr(self.materialize(), op)(*args)
    method.__name__ = op
    return method


for _op in ("add", "sub", "mul", "truediv", "floordiv", "pow", "matmul", "mod", "radd", "rsub", "rmul", "rtruediv",
            "rpow", "rmatmul", "neg", "pos", "abs", "lt", "le", "gt", "ge", "eq", "ne"):
    setattr(LazyWeights, f"__{_op}__", _delegate(f"__{_op}__"))
LazyWeights.__hash__ = None


class APAP:
    """As-Projective-As-Possible moving-DLT engine (GPU).  Constructor and method
    signatures follow reference apap.py:21-217."""

    def __init__(self, gamma, sigma, final_size, offset, device=-1, ctx=None):
        self.gamma = gamma
        self.sigma = sigma
        self.final_width, self.final_height = final_size
        self.offset_x, self.offset_y = offset
        self.device = device
        self.ctx = ctx          # a _native.Context (solver options, profiling); None = the defaults

    # ---- once-per-pair helpers: numpy, same arithmetic as the reference -------------
    @staticmethod
    def getNormalize2DPts(point):
        """Similarity that moves the centroid to the origin and the mean distance to
        sqrt(2); returns ``(t, t . point)``.  Reference apap.py:35-59."""
        count = point.shape[0]
        centre = np.mean(point, axis=0)
        shifted = point - centre
        mean_dist = np.mean(np.sqrt(np.sum(np.square(shifted), axis=1)))
        scale = np.sqrt(2) / (mean_dist + 1e-8)
        t = np.array([[scale, 0, -scale * centre[0]],
                      [0, scale, -scale * centre[1]],
                      [0, 0, 1]], dtype=np.float32)
        homog = np.column_stack((point, np.ones(count, dtype=np.float32)))
        return t, t.dot(homog.T).T[:, :2]

    @staticmethod
    def getConditionerFromPts(point):
        """Per-axis scaling to sample standard deviation sqrt(2).  Reference
        apap.py:63-89."""
        count = point.shape[0]
        mean_x, mean_y = np.mean(point, axis=0)
        std = np.std(point, axis=0)
        std_x, std_y = np.sqrt(std * std * count / (count - 1))
        std_x = std_x + (std_x == 0)
        std_y = std_y + (std_y == 0)
        nx, ny = np.sqrt(2) / std_x, np.sqrt(2) / std_y
        return np.array([[nx, 0, -nx * mean_x], [0, ny, -ny * mean_y], [0, 0, 1]], dtype=np.float32)

    @staticmethod
    def point_normalize(nf, c):
        """Apply the diagonal + translation of ``c`` to every point.  Reference
        apap.py:92-100."""
        cf = np.zeros_like(nf)
        cf[:, 0] = nf[:, 0] * c[0, 0] + c[0, 2]
        cf[:, 1] = nf[:, 1] * c[1, 1] + c[1, 2]
        return cf

    @staticmethod
    def matrix_generate(sample_n, cf1, cf2):
        """The ``2 sample_n x 9`` float32 DLT matrix.  Reference apap.py:103-119."""
        return _native.host_dlt_rows(np.asarray(cf1)[:sample_n], np.asarray(cf2)[:sample_n])

    @staticmethod
    def warp_coordinate_estimate(pt, homography):
        """``homography @ pt`` normalised by its third component.  Reference
        apap.py:172-184."""
        target = homography @ pt
        target /= target[2]
        return target

    # ---- hot loop 1 ------------------------------------------------------------------
    def local_homography(self, src_point, dst_point, vertices, return_weights=True):
        """Per-cell weighted DLT.  Returns ``(H, W)`` like reference apap.py:121-169:
        ``H`` float32 ``(rows, cols, 3, 3)``, ``W`` float64 ``(rows, cols, n)``.

        ``W`` is a :class:`LazyWeights`: it behaves like the reference's array (shape, dtype, indexing,
        ``np.asarray``, arithmetic) but its ``8 n`` bytes per cell are only computed and copied when looked at -
        the reference's own caller never does (apap.py:242), and an unmodified caller should not pay 640 MB of
        HBM and PCIe traffic per pair for it.  ``return_weights="eager"`` computes the ndarray in the same call
        (what the reference does); ``return_weights=False`` returns ``None``."""
        H, W = _native.local_homography(src_point, dst_point, vertices, self.gamma, self.sigma,
                                        want_weights=(return_weights == "eager"), device=self.device, ctx=self.ctx)
        if return_weights and not (isinstance(return_weights, str) and return_weights == "eager"):      # any truthy value but "eager"
            W = LazyWeights(src_point, vertices, self.gamma, self.sigma, device=self.device, ctx=self.ctx)
        return H, W

    # ---- hot loop 2 ------------------------------------------------------------------
    def local_warp(self, ori_img, local_homography, mesh, progress=False):
        """Backward warp of ``ori_img`` onto the canvas.  Reference apap.py:186-217.

        Like the reference, the per-cell inverses are written back INTO
        ``local_homography`` (apap.py:201-203) when it is a writable float32 array.
        ``progress`` is accepted for signature compatibility; one kernel launch has no
        rows to report."""
        mesh_w, mesh_h = mesh
        ori_h, ori_w, _ = ori_img.shape
        mesh_n, pt_size, _, _ = local_homography.shape
        print("Inverse solving started.")        # the reference prints these two lines (apap.py:200,204);
        warped, hinv = _native.local_warp(ori_img, local_homography, mesh_w, mesh_h, self.final_width,
                                          self.final_height, self.offset_x, self.offset_y,
                                          want_inverse=True, device=self.device, ctx=self.ctx)
        print("Inverse solving completed.")      # here the inverses come from the same native call as the warp
        if isinstance(local_homography, np.ndarray) and local_homography.flags.writeable:
            local_homography[...] = hinv
        return warped


    def local_stitch(self, ori_img, center_img, local_homography, mesh):
        """The blend the reference's ``__main__`` keeps commented out (apap.py:258-262) as
        one fused pass: ``uniform_blend(local_warp(ori_img, H, mesh), paste(center_img))``
        with the centre image pasted at ``(offset_x, offset_y)``.  Does not mutate
        ``local_homography``."""
        mesh_w, mesh_h = mesh
        out, _ = _native.local_stitch(ori_img, center_img, local_homography, mesh_w, mesh_h, self.final_width,
                                      self.final_height, self.offset_x, self.offset_y, device=self.device, ctx=self.ctx)
        return out


# ------------------------------------------------------------------------------------
# CLI: apap.py:220-265
# ------------------------------------------------------------------------------------
def save2mat(path, arr, name="sift_feature", prefix="./output/"):
    """Reference utils.py:68-70."""
    from .utils import save2mat as _save
    return _save(path, arr, name=name, prefix=prefix)


def read_config(path):
    """``key = value`` lines, the syntax of the reference's configs/case*.txt
    (options.py:12-13 reads them with configargparse).  Only ``mesh_size``, ``gamma``
    and ``sigma`` concern this path; the spectral-matching keys are ignored."""
    out = {}
    with open(path) as fh:
        for line in fh:
            line = line.split("#", 1)[0].strip()
            if "=" in line:
                k, v = (s.strip() for s in line.split("=", 1))
                out[k] = v
    return out


def run_pair(src, dst, H_global, other_shape, center_shape, mesh_size=100, gamma=0.5, sigma=100,
             other_img=None, center_img=None, device=-1, pipeline=None):
    """Body of the reference's ``__main__`` between loading and saving
    (apap.py:238-264).  Returns ``(H_flat (m*m, 9) float64, canvas or None)``; the canvas
    is the warped other image, or - when ``center_img`` is given - the blended stitch
    of apap.py:258-262.  One resident pass (:class:`cvx_proj_amd.pipeline.Pipeline`): the H grid never
    leaves HBM between the solve, the output stage and the warp."""
    from .pipeline import Pipeline
    pipe = pipeline if pipeline is not None else Pipeline(device=device)
    return pipe.run_pair(src, dst, H_global, other_shape, center_shape, mesh_size, gamma, sigma,
                         other_img=other_img, center_img=center_img)


def run_pair_by_calls(src, dst, H_global, other_shape, center_shape, mesh_size=100, gamma=0.5, sigma=100,
                      other_img=None, center_img=None, device=-1):
    """The same through the mirror class, call by call, as the reference's ``__main__`` is written: numpy in and
    out of every stage (the H grid crosses PCIe three times).  Kept as the yardstick of :func:`run_pair`."""

    class _S:
        def __init__(self, shape):
            self.shape = shape

    fw, fh, ox, oy = (int(v) for v in final_size(_S(center_shape), _S(other_shape), H_global))
    mesh = get_mesh((fw, fh), mesh_size + 1)
    vertices = get_vertice((fw, fh), mesh_size, (ox, oy))
    eng = APAP(gamma, sigma, [fw, fh], [ox, oy], device=device)
    H, _ = eng.local_homography(src, dst, vertices, return_weights=False)
    if other_img is not None and center_img is not None:
        warped = eng.local_stitch(other_img, center_img, H, mesh)
    else:
        warped = eng.local_warp(other_img, H.copy(), mesh) if other_img is not None else None
    flat = _native.invert_normalize_flatten(H, device=device)   # apap.py:250-264
    return flat, warped


def _parse_jobs(a, ap):
    """The (case_idx, img_idx) pairs of one invocation: the two positionals of apap.py:225-232, or the loop of
    ``--cases a-b --imgs i,j,...`` (the pattern of the reference's run_all.sh:4-31: 4 cases x pictures 1 2 4 5 in ONE process -
    one start-up of the interpreter, the runtime and the code objects for all of them)."""
    if a.cases is None and a.imgs is None:
        return [(a.case_idx, a.img_idx)]
    try:
        lo, _, hi = (a.cases or str(a.case_idx)).partition("-")
        cases = range(int(lo), int(hi or lo) + 1)
        imgs = [int(v) for v in (a.imgs or str(a.img_idx)).split(",")]
    except ValueError:
        ap.error("--cases takes a or a-b, --imgs a comma-separated list of picture indices")
    return [(c, i) for c in cases for i in imgs]


def main(argv=None):
    import argparse
    import time
    t_start = time.perf_counter()
    ap = argparse.ArgumentParser(prog="cvx_proj_amd.apap", description=__doc__.split("\n\n")[0])
    ap.add_argument("case_idx", nargs="?", type=int, default=1)
    ap.add_argument("img_idx", nargs="?", type=int, default=1)
    ap.add_argument("--cases", help="a or a-b: loop over these cases in one process (with --imgs; run_all.sh's pattern)")
    ap.add_argument("--imgs", help="comma-separated picture indices to loop over, e.g. 1,2,4,5")
    ap.add_argument("--pair", help=".npz with src, dst (n,2), H (3,3), other_shape, center_shape[, other_img]")
    ap.add_argument("--synth", help="use a synthetic configuration of cvx_proj_amd.synth (C1..C5)")
    ap.add_argument("--data-root", help="the reference's own flow (apap.py:236-238): read case{c}/scat/img_haze{i}.png and "
                                        "case{c}/keypoints.mat under this directory (the reference's is ../diff_1/raw_data), "
                                        "equalise, RANSAC seed homography, then the hot path")
    ap.add_argument("--config", help="key = value file; mesh_size / gamma / sigma are read")
    ap.add_argument("--mesh-size", type=int)
    ap.add_argument("--gamma", type=float)
    ap.add_argument("--sigma", type=float)
    ap.add_argument("--out-prefix", default="../diff_1/results/")
    ap.add_argument("--warp", help="also run local_warp and save the canvas to this .npy")
    ap.add_argument("--stitch", help="run the fused warp + blend with the centre image (the reference's "
                                     "commented-out tail, apap.py:258-262) and save the canvas to this .npy")
    ap.add_argument("--device", type=int, default=-1)
    ap.add_argument("--resident", action="store_true",
                    help="one resident pass per pair through cvx_proj_amd.pipeline (imports torch: ~2 s more start-up, ~0.3 ms less "
                         "per pair); default: the host-buffer entry points of the C ABI, no torch in the process")
    ap.add_argument("--timing", action="store_true", help="print the stages' wall times as one JSON line on stderr")
    a = ap.parse_args(argv)

    par = {"mesh_size": 100, "gamma": 0.5, "sigma": 100.0}     # apap.py:221-223
    if a.config:
        cfg = read_config(a.config)
        for k, cast in (("mesh_size", int), ("gamma", float), ("sigma", float)):
            if k in cfg:
                par[k] = cast(cfg[k])
    for k in par:
        v = getattr(a, k)
        if v is not None:
            par[k] = v
    if not (a.pair or a.data_root or a.synth):
        ap.error("the reference's dataset (../diff_1) is not distributed: give --data-root, --pair or --synth")
    jobs = _parse_jobs(a, ap)
    stages = {"imports_ms": (time.perf_counter() - t_start) * 1e3}

    pipeline = None
    t0 = time.perf_counter()
    if a.resident:
        import torch  # noqa: F401  (before the library is loaded: one HIP runtime per process, _native.lib)
        from .pipeline import Pipeline
        pipeline = Pipeline(device=a.device)
    else:
        if _native.lib().apap_device_count() < 1:       # loads the library, initialises the runtime
            raise _native.ApapError(_native.ERR_NO_DEVICE, "no HIP device; there is no CPU fallback")
    stages["runtime_init_ms"] = (time.perf_counter() - t0) * 1e3
    stages["pairs"] = []

    for case_idx, img_idx in jobs:
        t0 = time.perf_counter()
        mesh_size = par["mesh_size"]
        if a.pair:
            z = np.load(a.pair)
            src, dst, Hg = z["src"], z["dst"], z["H"]
            other_shape, center_shape = tuple(z["other_shape"]), tuple(z["center_shape"])
            other_img = z["other_img"] if ((a.warp or a.stitch) and "other_img" in z) else None
            center_img = z["center_img"] if (a.stitch and "center_img" in z) else None
        elif a.data_root:
            from .baseline_stitch_test import CENTER_PIC_ID, visualize_feature_pairs
            from .utils import equalize_hist, get_no_scat_img, get_path, imread
            shapes = []
            for idx in (CENTER_PIC_ID, img_idx):         # visualize_equalized_hist (apap.py:236-237): the reference shows the
                path = get_path(case_idx, idx, root=a.data_root)        # pictures and goes on with their shape
                img = imread(path)
                if img is None:
                    raise FileNotFoundError(path)
                if pipeline is not None:
                    pipeline.equalize(img, name=f"eq{idx}")
                else:
                    equalize_hist(img, device=a.device)
                shapes.append(img.shape)
            center_shape, other_shape = shapes
            src, dst, Hg = visualize_feature_pairs(None, None, case_idx=case_idx, pic_id=img_idx, swap=True,
                                                   root=a.data_root, device=a.device)
            other_img = center_img = None
            if a.warp or a.stitch:      # the blend uses the haze-free pictures (apap.py:245,258-262)
                center_img, other_img = get_no_scat_img(case_idx, img_idx, CENTER_PIC_ID, root=a.data_root)
                if not a.stitch:
                    center_img = None
        else:
            from .synth import CONFIGS, synth_pair
            w, h, n, m, seed = CONFIGS[a.synth]
            if a.mesh_size is None and not a.config:
                mesh_size = m
            # in a loop every (case, picture) is a pair of its own: the first one is the configuration's own seed
            seed += (case_idx - jobs[0][0]) * 16 + (img_idx - jobs[0][1])
            p = synth_pair(w, h, n, mesh_size, seed, with_image=bool(a.warp or a.stitch))
            src, dst, Hg, other_shape, center_shape, other_img = p.src, p.dst, p.Hg, p.shape, p.shape, p.img
            center_img = (np.random.default_rng(seed + 1).integers(0, 256, p.shape, dtype=np.uint8) if a.stitch else None)
        t1 = time.perf_counter()
        if pipeline is not None:
            flat, warped = run_pair(src, dst, Hg, other_shape, center_shape, mesh_size, par["gamma"], par["sigma"],
                                    other_img=other_img, center_img=center_img, device=a.device, pipeline=pipeline)
        else:
            flat, warped = run_pair_by_calls(src, dst, Hg, other_shape, center_shape, mesh_size, par["gamma"], par["sigma"],
                                             other_img=other_img, center_img=center_img, device=a.device)
        t2 = time.perf_counter()
        print(f"local_homography shape: {(mesh_size, mesh_size, 3, 3)}")
        out_dir = f"{a.out_prefix}case{case_idx}"
        os.makedirs(out_dir, exist_ok=True)
        save2mat(f"case{case_idx}/H3{img_idx}_apap", flat, name="H", prefix=a.out_prefix)
        if (a.warp or a.stitch) and warped is not None:
            out = a.stitch or a.warp
            if len(jobs) > 1:
                root, ext = os.path.splitext(out)
                out = f"{root}_case{case_idx}_{img_idx}{ext}"
            np.save(out, warped)
        stages["pairs"].append({"case": case_idx, "img": img_idx, "inputs_ms": (t1 - t0) * 1e3, "compute_ms": (t2 - t1) * 1e3,
                                "save_ms": (time.perf_counter() - t2) * 1e3})
    stages["total_ms"] = (time.perf_counter() - t_start) * 1e3
    if a.timing:
        import json
        print(json.dumps(stages), file=sys.stderr, flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
