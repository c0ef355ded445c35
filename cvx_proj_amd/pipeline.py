"""The reference's ``apap.py __main__`` (apap.py:234-265) as ONE resident pass over the GPU.

The mirror class of :mod:`cvx_proj_amd.apap` keeps the reference's call-by-call surface: every call
takes numpy buffers, copies them up, runs its kernels, copies the result down.  Chained the way
``__main__`` chains them, the H grid crosses PCIe three times (down after the solve, up again for the warp,
up a third time for the output stage) and every stage ends in a synchronisation.  A :class:`Pipeline`
keeps what the stages hand to each other in HBM - keypoint table, H grid, source image, canvas - enqueues

    (equalise ->) (RANSAC seed ->) solve -> output stage -> warp / stitch

on one HIP stream through the resident ``*_device`` entry points of ``libapap_hip.so`` and copies back
once, at the end: the ``(m*m, 9)`` float64 array that goes into ``H3{i}_apap.mat`` and, when asked for,
the canvas.  Same kernels, same bits as the chain of calls (``tests/test_pipeline.py``).

Only the seed homography has to visit the host in between: ``final_size`` (apap_utils.py:40-73) turns it
into the canvas geometry, mesh and vertices, which size everything after it.

torch is used for device memory and the stream only.  No CPU fallback.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _native
from .apap_utils import final_size, get_mesh, get_vertice

__all__ = ["Pipeline"]


class _Shape:
    def __init__(self, shape):
        self.shape = tuple(shape)


class Pipeline:
    """Resident APAP pass.  ``device``: HIP device index (-1 = current); ``ctx``: a ``_native.Context``."""

    def __init__(self, device=-1, ctx=None):
        import sys
        if _native._lib is not None and "torch" not in sys.modules:
            raise _native.ApapError(_native.ERR_HIP, "cvx_proj_amd's library was loaded before torch was imported: this process already "
                                                     "runs /opt/rocm's HIP runtime, torch would not see the GPU through it.  Import torch "
                                                     "before the first call into cvx_proj_amd (or set APAP_HIP_PRELOAD_TORCH=1)")
        import torch
        if not torch.cuda.is_available():
            raise _native.ApapError(_native.ERR_NO_DEVICE, "Pipeline needs a HIP device; there is no CPU fallback")
        self.torch = torch
        self.dev = torch.device("cuda", torch.cuda.current_device() if device < 0 else device)
        self.ctx = ctx
        self._h = _native._h(ctx)
        self._buf = {}              # name -> device tensor, grown on demand and reused between pairs
        self.timeline = {}          # milliseconds of the last run_pair, by stage (host clock, after the final sync)
        self._plans = {}            # canvas geometry -> WarpPlan (lookup tables built once per geometry)
        self._pinned = []           # page-locked host buffers handed out by pinned_array()
        self._side = torch.cuda.Stream(self.dev)          # download of the .mat array beside the warp
        self.trace = False          # True: HIP events at the stage boundaries of run_pair -> self.device_marks (us from the first enqueue)
        self.device_marks = []

    # ------------------------------------------------------------------ device memory
    def _get(self, name, shape, dtype):
        t = self._buf.get(name)
        n = int(np.prod(shape))
        if t is None or t.dtype != dtype or t.numel() < n:
            t = self.torch.empty(max(n, 1), dtype=dtype, device=self.dev)
            self._buf[name] = t
        return t[:n].view(*shape)

    def _up(self, name, array, dtype):
        a = np.ascontiguousarray(array)
        t = self._get(name, a.shape, dtype)
        t.copy_(self.torch.from_numpy(a), non_blocking=True)
        return t

    def _down(self, t):
        """Device tensor -> a new numpy array, copied straight into numpy's own allocation (a `.cpu()` tensor is a
        fresh mapping every time: a 27 MB canvas then pays ~2 ms of page faults before the copy starts)."""
        a = np.empty(tuple(t.shape), dtype={self.torch.float64: np.float64, self.torch.float32: np.float32,
                                            self.torch.uint8: np.uint8}[t.dtype])
        self.torch.from_numpy(a).copy_(t)
        return a

    def _stream(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def pinned_array(self, shape, dtype=np.uint8):
        """A numpy array in page-locked host memory (owned by this pipeline).  A caller that reads its images INTO such arrays
        (``np.copyto(buf, cv.imread(...))``, or a decoder writing in place) and passes them to ``run_pair`` gets asynchronous
        copies: the image goes up beside the host set-up and the solve instead of after them, and ``canvas_out=`` receives
        the canvas without the pin-and-copy detour of pageable memory.  Ordinary numpy arrays work as before."""
        torch = self.torch
        t = torch.empty(tuple(shape), dtype={np.dtype(np.uint8): torch.uint8, np.dtype(np.float32): torch.float32,
                                             np.dtype(np.float64): torch.float64}[np.dtype(dtype)], pin_memory=True)
        self._pinned.append(t)
        return t.numpy()

    def _staging(self, n):
        """`n` float64 of page-locked staging memory (one block, reused by every pass; a pass ends synchronised)."""
        t = getattr(self, "_stage_t", None)
        if t is None or t.numel() < n:
            t = self._stage_t = self.torch.empty(max(n, 1 << 16), dtype=self.torch.float64, pin_memory=True)
        return t.numpy()

    def _is_pinned(self, a):
        try:
            return isinstance(a, np.ndarray) and a.flags.c_contiguous and a.dtype == np.uint8 and self.torch.from_numpy(a).is_pinned()
        except (TypeError, ValueError, RuntimeError):
            return False

    # ------------------------------------------------------------------ stages
    def equalize(self, img, name="eq", fetch=False):
        """Per-channel ``cv.equalizeHist`` (utils.py:85-91) of an (h, w, c) uint8 image; the result stays on the
        device (``fetch=True`` also returns it as numpy: the reference only ever looks at its shape and shows it)."""
        lib = _native.lib()
        img = np.ascontiguousarray(img, dtype=np.uint8)
        ch = 1 if img.ndim == 2 else img.shape[2]
        d_in = self._up(name + "_in", img, self.torch.uint8)
        d_out = self._get(name + "_out", img.shape, self.torch.uint8)
        nbytes = lib.apap_equalize_workspace_bytes(ch)
        work = self._buf.get("eq_work")
        if work is None or work.numel() < nbytes:
            work = self._buf["eq_work"] = self.torch.zeros(nbytes, dtype=self.torch.uint8, device=self.dev)   # zero on entry, zero on return
        _native.check(lib.apap_equalize_hist_device(self._h, d_in.data_ptr(), img.shape[0], img.shape[1], ch, d_out.data_ptr(),
                                                    work.data_ptr(), nbytes, self._stream()))
        return self._down(d_out) if fetch else d_out

    def seed_homography(self, src_pts, dst_pts, thresh=5.0):
        """``cv.findHomography(src, dst, cv.RANSAC, thresh)``'s contract (baseline_stitch_test.py:42).  The
        inlier mask and the 3 x 3 model come to the host: the canvas geometry is computed from them."""
        return _native.find_homography_ransac(src_pts, dst_pts, thresh, device=self.dev.index, ctx=self.ctx)

    def _plan(self, mesh, rows, cols, fw, fh, ox, oy):
        """The warp workspace of one mesh / canvas geometry: the canvas row / column -> cell tables are built when the geometry is
        first seen (they do not depend on H) and kept - a CLI run over the pictures of one case, or a caller streaming pairs of one
        size, meets a handful of geometries."""
        from .dist import WarpPlan
        key = (rows, cols, fw, fh, ox, oy, mesh.tobytes())
        plan = self._plans.get(key)
        if plan is None:
            if len(self._plans) >= 8:
                self._plans.pop(next(iter(self._plans)))
            plan = self._plans[key] = WarpPlan(mesh, (rows, cols), fw, fh, ox, oy, self.dev, batch=1, ctx=self.ctx)
            plan.vertices = None        # the mesh's sample points on the device, kept with the geometry
        return plan

    def run_pair(self, src, dst, H_global, other_shape, center_shape, mesh_size=100, gamma=0.5, sigma=100,
                 other_img=None, center_img=None, want_grid=False, canvas_out=None):
        """Body of the reference's ``__main__`` between loading and saving (apap.py:238-264): returns
        ``(H_flat (m*m, 9) float64, canvas or None)`` (and the float32 H grid with ``want_grid``); the canvas is
        the warped other image or, with ``center_img``, the blended stitch of apap.py:258-262.

        What overlaps (round 4): with ordinary numpy images the solve is enqueued FIRST and the 25 MB source image goes up while it
        runs (a copy from pageable memory blocks the host, not the GPU; a helper thread for the upload was measured: no gain).
        With the images in page-locked arrays (``pinned_array``) the image upload starts before anything else on the side stream
        and the host set-up, the small uploads and the solve run beside it: every copy of such a pass is a page-locked one -
        keypoint table and de-normalisation are built straight into a staging block, the ``.mat`` array and the status word come
        down into staging - because ONE pageable copy on the side stream makes the small uploads of every later pass wait for
        the whole image upload (measured: profiles/r04_pipeline_overlap.txt; its script at git tag r05-hooks; 1.33 -> 1.04 ms).
        The solve's tail leaves every cell warp ready, so the warp is the gather kernel alone on tables whose geometry half
        (and the mesh's vertices) were built when this canvas geometry was first seen; the canvas download is the one copy
        left on the critical path (0.5 ms of the 1.04).  ``self.trace = True`` records HIP events at the stage boundaries
        (``self.device_marks``)."""
        import time
        torch, lib = self.torch, _native.lib()
        t0 = time.perf_counter()
        early = None
        trace = self.trace
        pinned_in = other_img is not None and self._is_pinned(other_img) and (center_img is None or self._is_pinned(center_img))

        def start_image_upload():
            # page-locked images: truly asynchronous copies on the side stream
            with torch.cuda.device(self.dev), torch.cuda.stream(self._side):
                d_img = self._get("img", other_img.shape, torch.uint8)
                d_img.copy_(torch.from_numpy(other_img), non_blocking=True)
                d_cen = None
                if center_img is not None:
                    d_cen = self._get("center", center_img.shape, torch.uint8)
                    d_cen.copy_(torch.from_numpy(center_img), non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._side)
            return (d_img, d_cen, ev)

        marks = []

        def mark(name, stream=None):
            if trace:
                e = torch.cuda.Event(enable_timing=True)
                e.record(stream if stream is not None else torch.cuda.current_stream(self.dev))
                marks.append((name, e))

        mark("begin", self._side)
        if pinned_in:
            early = start_image_upload()
            mark("image up (side)", self._side)
        fw, fh, ox, oy = (int(v) for v in final_size(_Shape(center_shape), _Shape(other_shape), H_global))
        mesh = get_mesh((fw, fh), mesh_size + 1)
        rows = cols = int(mesh_size)
        cells = rows * cols
        plan = None
        with torch.cuda.device(self.dev):
            if other_img is not None and mesh.shape[1] <= 4096:
                plan = self._plan(mesh, rows, cols, fw, fh, ox, oy)     # lookup tables + vertices of this geometry: built once
        vertices = None if plan is not None and plan.vertices is not None else get_vertice((fw, fh), mesh_size, (ox, oy))
        q = _native.host_prepare(src, dst)                                  # apap.py:132-140 in C
        # keypoint table and de-normalisation written straight into ONE page-locked staging block: a single asynchronous copy.
        # (A copy from pageable memory, however small, queues behind the 25 MB image copy in the runtime: measured 350 us of
        # waiting, the solve starting when the upload had finished - profiles/r04_pipeline_overlap.txt.)
        n = int(np.asarray(src).shape[0])
        nt, nd = n * _native.TABLE_STRIDE, _native.DENORM_DOUBLES
        stage = self._staging(nt + nd)
        table = _native.host_build_table(src, q["cf1"], q["cf2"], out=stage[:nt].reshape(n, _native.TABLE_STRIDE),
                                         moments=self.ctx.get("moments") if self.ctx is not None else 30)
        _native.host_build_denorm(q["iC2"], q["C1"], q["iN2"], q["N1"], out=stage[nt:nt + nd])
        t1 = time.perf_counter()
        with torch.cuda.device(self.dev):
            stream = self._stream()
            main = torch.cuda.current_stream(self.dev)
            mark("main: first enqueue")
            d_stage = self._get("stage", (nt + nd,), torch.float64)
            d_stage.copy_(self._stage_t[:nt + nd], non_blocking=True)
            d_table, d_den = d_stage[:nt].view(n, _native.TABLE_STRIDE), d_stage[nt:]
            if vertices is None:
                d_vert = plan.vertices
            else:
                d_vert = self._up("vertices", vertices.reshape(-1, 2), torch.float64)
                if plan is not None:
                    plan.vertices = d_vert.clone()
            d_H = self._get("H", (cells, 9), torch.float32)
            nb = max(lib.apap_solve_workspace_bytes(self._h, n, cells), 256)
            d_work = self._get("solve_work", (nb,), torch.uint8)
            if plan is not None:
                # the solve that leaves every cell warp ready in the plan's workspace (apap_solve_warp_batch_device)
                plan.begin()        # this pair's status bits; the geometry phase's are kept in plan.geo_status
                mark("small uploads done")
                plan.solve(d_table, d_den, d_vert, float(gamma), float(sigma), out=d_H, work=d_work)
                mark("solve done")
            else:
                _native.check(lib.apap_solve_device(self._h, d_table.data_ptr(), n, d_vert.data_ptr(), cells, float(gamma), float(sigma),
                                                    d_den.data_ptr(), d_H.data_ptr(), d_work.data_ptr(), nb, stream))
            # output stage apap.py:250-264 on the resident grid
            d_flat = self._get("flat", (cells, 9), torch.float64)
            d_status = plan.status if plan is not None else self._get("status", (1,), torch.int32)     # one word for every stage
            if plan is None:
                d_status.zero_()
            _native.check(lib.apap_flatten_device(self._h, d_H.data_ptr(), cells, d_flat.data_ptr(), d_status.data_ptr(), stream))
            flat_ready = torch.cuda.Event()
            flat_ready.record(main)
            d_out = None
            if other_img is not None:
                img = np.ascontiguousarray(other_img, dtype=np.uint8)
                if early is not None:
                    d_img, d_cen, ev = early
                    main.wait_event(ev)
                    mark("image here")
                    cen = center_img
                else:
                    d_img = self._up("img", img, torch.uint8)       # blocks the host while the GPU solves
                    d_cen = None
                    if center_img is not None:
                        cen = np.ascontiguousarray(center_img, dtype=np.uint8)
                        d_cen = self._up("center", cen, torch.uint8)
                d_out = self._get("canvas", (fh, fw, 3), torch.uint8)
                if plan is not None:
                    plan.gather(d_img, out=d_out.view(1, fh, fw, 3), centers=d_cen)
                    mark("warp done")
                else:
                    d_mw = self._up("mesh_w", mesh[0], torch.float64)
                    d_mh = self._up("mesh_h", mesh[1], torch.float64)
                    wb = lib.apap_warp_workspace_bytes(rows, cols, fw, fh)
                    d_ww = self._get("warp_work", (wb,), torch.uint8)
                    _native.check(lib.apap_warp_batch_device(self._h, d_img.data_ptr(), 0, img.shape[0], img.shape[1],
                                                             None if d_cen is None else d_cen.data_ptr(), 0,
                                                             0 if d_cen is None else cen.shape[0], 0 if d_cen is None else cen.shape[1],
                                                             d_H.data_ptr(), rows, cols, d_mw.data_ptr(), mesh.shape[1], d_mh.data_ptr(),
                                                             mesh.shape[1], fw, fh, ox, oy, 0, fh, d_out.data_ptr(), 0, None, 1,
                                                             _native.WARP_ALL, d_ww.data_ptr(), wb, d_status.data_ptr(), stream))
            # the status word follows the last kernel into page-locked memory: read after the final synchronisation, no copy of its own
            if getattr(self, "_status_host", None) is None:
                self._status_host = torch.zeros(1, dtype=torch.int32, pin_memory=True)
            self._status_host.copy_(d_status, non_blocking=True)
            t2 = time.perf_counter()
            # the .mat array comes down on the side stream beside the warp kernel, the canvas after it
            flat = flat_done = None
            with torch.cuda.stream(self._side):
                self._side.wait_event(flat_ready)
                if pinned_in:
                    # page-locked staging, copied into the result while the canvas comes down.  NOT a copy into pageable memory:
                    # one pageable copy on the side stream and, from the next pass on, the small uploads on the main stream wait
                    # for the whole image upload on the side stream (476 instead of 55 us, profiles/r04_pipeline_overlap.txt)
                    if getattr(self, "_flat_host", None) is None or self._flat_host.numel() < cells * 9:
                        self._flat_host = torch.empty(cells * 9, dtype=torch.float64, pin_memory=True)
                    self._flat_host[:cells * 9].copy_(d_flat.view(-1), non_blocking=True)
                    flat_done = torch.cuda.Event()
                    flat_done.record(self._side)
                else:
                    flat = self._down(d_flat)
            if d_out is not None and canvas_out is not None:
                if canvas_out.shape != (fh, fw, 3) or canvas_out.dtype != np.uint8 or not canvas_out.flags.c_contiguous:
                    raise ValueError(f"canvas_out must be a contiguous uint8 array of shape {(fh, fw, 3)}")
                torch.from_numpy(canvas_out).copy_(d_out, non_blocking=True)
                mark("canvas down")
                if flat_done is not None:
                    flat_done.synchronize()
                    flat = self._flat_host[:cells * 9].numpy().reshape(cells, 9).copy()
                main.synchronize()
                canvas = canvas_out
            else:
                if flat_done is not None:
                    flat_done.synchronize()
                    flat = self._flat_host[:cells * 9].numpy().reshape(cells, 9).copy()
                canvas = self._down(d_out) if d_out is not None else None      # (synchronises the main stream)
            main.synchronize()
            status = int(self._status_host[0]) | (plan.geo_status if plan is not None else 0)
            grid = self._down(d_H).reshape(rows, cols, 3, 3) if want_grid else None
        t3 = time.perf_counter()
        if marks:
            torch.cuda.synchronize(self.dev)
            self.device_marks = [(n, marks[0][1].elapsed_time(e) * 1e3) for n, e in marks]
        self.timeline = {"host_setup_ms": (t1 - t0) * 1e3, "upload_and_enqueue_ms": (t2 - t1) * 1e3,
                         "sync_and_download_ms": (t3 - t2) * 1e3, "total_ms": (t3 - t0) * 1e3}
        if status & 1:
            raise _native.ApapSingularError(_native.ERR_SINGULAR, "Singular matrix")
        if status & 4:
            raise _native.ApapValueError(_native.ERR_INVALID_ARG, "warp workspace without lookup tables for this geometry")
        if status & 2:
            raise _native.ApapIndexError(_native.ERR_INDEX, "index 0 is out of bounds for axis 0 with size 0 (mesh edges do not "
                                                             "cover the canvas)")
        return (flat, canvas, grid) if want_grid else (flat, canvas)
