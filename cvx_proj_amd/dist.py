"""Multi-GPU driver for the APAP path: one process per GPU, ``torch.distributed``
(backend ``nccl`` = RCCL over xGMI on ROCm; ``gloo`` in the CPU tests).

The path shards in two ways:

* **cells of one pair** (config C4): every mesh cell is independent given the keypoint
  table.  Rank 0 prepares the table on its host and *broadcasts* it (n x 256 B - 1.3 MB at
  n = 5000); each rank solves a contiguous block of mesh rows in two launches; the *all-gather*
  of the first half of the H grid (36 B per cell - 5.8 MB at 400 x 400, 720 KB per rank: on fully
  connected xGMI that is one hop and latency-bound) runs while the second half is computed.
* **independent pairs** (config C5): pairs are dealt round-robin to the ranks; no
  collective on the data path, one gather of the H grids at the end.

The compute itself is injected (``solve_fn``) so that the partitioning and the
collectives can be tested on CPU ranks; the default is the HIP engine and there is no
CPU fallback in this module.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _native


def row_partition(rows, world):
    """Contiguous, near-equal blocks of mesh rows: ``[(start, stop)] * world``.  The first
    ``rows % world`` ranks get one row more; ranks beyond ``rows`` get empty blocks."""
    base, extra = divmod(rows, world)
    out, start = [], 0
    for r in range(world):
        size = base + (1 if r < extra else 0)
        out.append((start, start + size))
        start += size
    return out


def hip_solve(table, denorm, vertices, gamma, sigma, ctx=None, out=None, work=None):
    """Default ``solve_fn``: resident-data C-ABI call on the tensors' device.  ``out`` (>= cells x 9 float32)
    and ``work`` (uint8 scratch) are reused when given and large enough - a solver that runs every step keeps
    them - else allocated."""
    if not table.is_cuda:
        raise _native.ApapError(_native.ERR_NO_DEVICE, "hip_solve needs CUDA/HIP tensors; there is no CPU fallback")
    cells = vertices.shape[0]
    n = table.shape[0]
    H = out[:cells] if out is not None and out.shape[0] >= cells else torch.empty((cells, 9), dtype=torch.float32, device=table.device)
    if cells == 0:
        return H
    nbytes = max(_native.lib().apap_solve_workspace_bytes(_native._h(ctx), n, cells), 256)
    if work is None or work.numel() < nbytes:
        work = torch.empty(nbytes, dtype=torch.uint8, device=table.device)
    stream = torch.cuda.current_stream(table.device).cuda_stream
    _native.check(_native.lib().apap_solve_device(_native._h(ctx), table.data_ptr(), n, vertices.data_ptr(), cells, float(gamma),
                                                  float(sigma), denorm.data_ptr(), H.data_ptr(), work.data_ptr(),
                                                  nbytes, ctypes.c_void_p(stream)))
    return H


def hip_warp_rows(img, H, mesh_w, mesh_h, final_w, final_h, off_x, off_y, row_begin, row_count, out_band, shape, ctx=None,
                  work=None, status=None):
    """Default ``warp_fn``: resident-data C-ABI call warping canvas rows
    ``[row_begin, row_begin + row_count)`` into ``out_band``.  ``work`` / ``status`` are reused when given."""
    if not img.is_cuda:
        raise _native.ApapError(_native.ERR_NO_DEVICE, "hip_warp_rows needs CUDA/HIP tensors; there is no CPU fallback")
    rows, cols = shape
    nbytes = _native.lib().apap_warp_workspace_bytes(rows, cols, final_w, final_h)
    if work is None or work.numel() < nbytes:
        work = torch.empty(nbytes, dtype=torch.uint8, device=img.device)
    if status is None:
        status = torch.zeros(1, dtype=torch.int32, device=img.device)
    stream = torch.cuda.current_stream(img.device).cuda_stream
    _native.check(_native.lib().apap_warp_rows_device(
        _native._h(ctx), img.data_ptr(), img.shape[0], img.shape[1], H.data_ptr(), rows, cols, mesh_w.data_ptr(), mesh_w.numel(),
        mesh_h.data_ptr(), mesh_h.numel(), final_w, final_h, off_x, off_y, row_begin, row_count, out_band.data_ptr(),
        work.data_ptr(), nbytes, status.data_ptr(), ctypes.c_void_p(stream)))
    return status


def hip_solve_batch(tables, denorms, vertices, gamma, sigma, ctx=None):
    """Several pairs (equal keypoint and cell counts, one shared mesh) in ONE launch:
    ``tables`` (B, n, 32), ``denorms`` (B, 36), ``vertices`` (cells, 2) -> H (B, cells, 9)."""
    if not tables.is_cuda:
        raise _native.ApapError(_native.ERR_NO_DEVICE, "hip_solve_batch needs CUDA/HIP tensors; there is no CPU fallback")
    batch, n = tables.shape[0], tables.shape[1]
    cells = vertices.shape[0]
    H = torch.empty((batch, cells, 9), dtype=torch.float32, device=tables.device)
    nbytes = max(_native.lib().apap_solve_batch_workspace_bytes(_native._h(ctx), n, cells, batch), 256)
    work = torch.empty(nbytes, dtype=torch.uint8, device=tables.device)
    stream = torch.cuda.current_stream(tables.device).cuda_stream
    _native.check(_native.lib().apap_solve_batch_device(_native._h(ctx), tables.data_ptr(), n, vertices.data_ptr(), 0, cells,
                                                        float(gamma), float(sigma), denorms.data_ptr(), H.data_ptr(),
                                                        batch, work.data_ptr(), nbytes, ctypes.c_void_p(stream)))
    return H


def hip_warp_batch(imgs, H, mesh_w, mesh_h, final_w, final_h, off_x, off_y, shape, out=None, centers=None, ctx=None,
                   work=None, status=None, phases=_native.WARP_ALL, rows=None, hinv_out=None):
    """Backward warp of a BATCH of independent pairs in one set of launches (``apap_warp_batch_device``, grid.z = pair):
    ``imgs`` (B, h, w, 3) uint8 - or (h, w, 3): one image for every pair -, ``H`` (B, cells, 9) float32 (what
    ``hip_solve_batch`` returns), one set of edges, canvas size and offsets for all -> canvases (B, final_h, final_w, 3).
    ``centers`` (B, ch, cw, 3) or (ch, cw, 3): the fused stitch (warp + paste + uniform_blend).  ``phases``: which of
    geometry tables / per-cell set-up / gather run on ``work`` (a caller that keeps ``work`` runs the geometry once).
    ``rows`` = (row_begin, row_count): a band of every canvas; ``out`` is then (B, row_count, final_w, 3)."""
    if not H.is_cuda:
        raise _native.ApapError(_native.ERR_NO_DEVICE, "hip_warp_batch needs CUDA/HIP tensors; there is no CPU fallback")
    mrows, mcols = shape
    batch = H.shape[0]
    dev = H.device
    row_begin, row_count = (0, final_h) if rows is None else rows
    one_img = imgs.dim() == 3
    ih, iw = (imgs.shape[0], imgs.shape[1]) if one_img else (imgs.shape[1], imgs.shape[2])
    if out is None:
        out = torch.empty((batch, row_count, final_w, 3), dtype=torch.uint8, device=dev)
    nbytes = _native.lib().apap_warp_batch_workspace_bytes(mrows, mcols, final_w, final_h, batch)
    if work is None or work.numel() < nbytes:
        work = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    if status is None:
        status = torch.zeros(1, dtype=torch.int32, device=dev)
    c_ptr, c_stride, ch, cw = None, 0, 0, 0
    if centers is not None:
        one_c = centers.dim() == 3
        ch, cw = (centers.shape[0], centers.shape[1]) if one_c else (centers.shape[1], centers.shape[2])
        c_ptr, c_stride = centers.data_ptr(), 0 if one_c else ch * cw * 3
    stream = torch.cuda.current_stream(dev).cuda_stream
    _native.check(_native.lib().apap_warp_batch_device(
        _native._h(ctx), imgs.data_ptr(), 0 if one_img else ih * iw * 3, ih, iw, c_ptr, c_stride, ch, cw, H.data_ptr(), mrows, mcols,
        mesh_w.data_ptr(), mesh_w.numel(), mesh_h.data_ptr(), mesh_h.numel(), final_w, final_h, off_x, off_y, row_begin, row_count,
        out.data_ptr(), row_count * final_w * 3, None if hinv_out is None else hinv_out.data_ptr(), batch, int(phases),
        work.data_ptr(), work.numel(), status.data_ptr(), ctypes.c_void_p(stream)))
    return out, status


class WarpPlan:
    """One mesh / canvas geometry's warp workspace, kept between pairs (what the resident callers - ``Pipeline``, ``bench.py`` -
    hold): the canvas row / column -> cell tables are built ONCE, here (``APAP_WARP_GEOMETRY``: they depend on the edges,
    the canvas size and the offsets only); ``solve()`` is the per-cell solve whose tail leaves every cell warp ready in
    this workspace (``apap_solve_warp_batch_device``); ``cells()`` does that for a grid that came from elsewhere
    (``APAP_WARP_CELLS``); ``gather()`` is K3 alone (``APAP_WARP_GATHER``).  ``batch`` pairs share the geometry."""

    def __init__(self, mesh, shape, final_w, final_h, off_x, off_y, dev, batch=1, ctx=None):
        self.rows, self.cols = shape
        self.geo = (int(final_w), int(final_h), int(off_x), int(off_y))
        self.dev, self.batch, self.ctx = dev, int(batch), ctx
        self.mesh_w = torch.from_numpy(np.ascontiguousarray(mesh[0], dtype=np.float64)).to(dev)
        self.mesh_h = torch.from_numpy(np.ascontiguousarray(mesh[1], dtype=np.float64)).to(dev)
        self.nbytes = _native.lib().apap_warp_batch_workspace_bytes(self.rows, self.cols, self.geo[0], self.geo[1], self.batch)
        if not self.nbytes:
            raise ValueError("WarpPlan: bad geometry")
        self.work = torch.zeros(self.nbytes, dtype=torch.uint8, device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self._phase(_native.WARP_GEOMETRY)
        # The kernels only OR bits into the status word.  What the geometry phase found (bit 1: the edges do not cover the
        # canvas) holds for the plan's whole life and is kept apart; the word itself then serves one pair at a time
        # (begin() before a pair's phases, status_word() after them), so that one pair's singular grid is not every later pair's.
        self.geo_status = int(self.status.cpu()[0])
        self.status.zero_()

    def begin(self):
        """Clear the per-pair status bits (singular cell, unprepared workspace) before a pair's phases."""
        self.status.zero_()

    def status_word(self):
        """This pair's status bits together with the geometry phase's (synchronises)."""
        return int(self.status.cpu()[0]) | self.geo_status

    def _phase(self, phases, imgs=None, H=None, out=None, centers=None, rows=None, hinv_out=None):
        fw, fh, ox, oy = self.geo
        row_begin, row_count = (0, fh) if rows is None else rows
        i_ptr, i_stride, ih, iw = None, 0, 0, 0
        if imgs is not None:
            one = imgs.dim() == 3
            ih, iw = (imgs.shape[0], imgs.shape[1]) if one else (imgs.shape[1], imgs.shape[2])
            i_ptr, i_stride = imgs.data_ptr(), 0 if one else ih * iw * 3
        c_ptr, c_stride, ch, cw = None, 0, 0, 0
        if centers is not None:
            one_c = centers.dim() == 3
            ch, cw = (centers.shape[0], centers.shape[1]) if one_c else (centers.shape[1], centers.shape[2])
            c_ptr, c_stride = centers.data_ptr(), 0 if one_c else ch * cw * 3
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        _native.check(_native.lib().apap_warp_batch_device(
            _native._h(self.ctx), i_ptr, i_stride, ih, iw, c_ptr, c_stride, ch, cw, None if H is None else H.data_ptr(), self.rows,
            self.cols, self.mesh_w.data_ptr(), self.mesh_w.numel(), self.mesh_h.data_ptr(), self.mesh_h.numel(), fw, fh, ox, oy,
            row_begin, row_count, None if out is None else out.data_ptr(), row_count * fw * 3,
            None if hinv_out is None else hinv_out.data_ptr(), self.batch, int(phases), self.work.data_ptr(), self.nbytes,
            self.status.data_ptr(), ctypes.c_void_p(stream)))

    def solve(self, tables, denorms, vertices, gamma, sigma, out=None, work=None):
        """``tables`` (B, n, 32) or (n, 32), ``denorms`` (B, 36) or (36,), ``vertices`` (cells, 2) -> H (B * cells, 9) float32, and
        every cell's inverse / record / exact floats in this plan's workspace."""
        if not tables.is_cuda:
            raise _native.ApapError(_native.ERR_NO_DEVICE, "WarpPlan.solve needs CUDA/HIP tensors; there is no CPU fallback")
        n = tables.shape[-2]
        cells = self.rows * self.cols
        if vertices.shape[0] != cells:
            raise ValueError(f"WarpPlan.solve: {vertices.shape[0]} vertices for a {self.rows} x {self.cols} mesh")
        H = out if out is not None else torch.empty((self.batch * cells, 9), dtype=torch.float32, device=self.dev)
        lib = _native.lib()
        nb = max(lib.apap_solve_batch_workspace_bytes(_native._h(self.ctx), n, cells, self.batch), 256)
        if work is None or work.numel() < nb:
            work = torch.empty(nb, dtype=torch.uint8, device=self.dev)
        fw, fh, ox, oy = self.geo
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        _native.check(lib.apap_solve_warp_batch_device(
            _native._h(self.ctx), tables.data_ptr(), n, vertices.data_ptr(), 0, float(gamma), float(sigma), denorms.data_ptr(),
            H.data_ptr(), self.batch, work.data_ptr(), work.numel(), self.rows, self.cols, self.mesh_w.data_ptr(), self.mesh_w.numel(),
            self.mesh_h.data_ptr(), self.mesh_h.numel(), fw, fh, ox, oy, self.work.data_ptr(), self.nbytes, self.status.data_ptr(),
            ctypes.c_void_p(stream)))
        return H

    def cells(self, H, hinv_out=None):
        """Per-cell set-up from a grid that was not solved into this plan (``H`` (B * cells, 9) float32)."""
        self._phase(_native.WARP_CELLS, H=H, hinv_out=hinv_out)

    def gather(self, imgs, out=None, centers=None, rows=None):
        """K3: ``imgs`` (B, h, w, 3) or (h, w, 3) -> canvases (B, rows, final_w, 3)."""
        fw, fh, _, _ = self.geo
        n = fh if rows is None else rows[1]
        if out is None:
            out = torch.empty((self.batch, n, fw, 3), dtype=torch.uint8, device=self.dev)
        self._phase(_native.WARP_GATHER, imgs=imgs, out=out, centers=centers, rows=rows)
        return out


# Rehearsals only (tests/_dist_gpu_ranks.py, tests/test_dist_gloo.py): True = a process group of ONE rank does not take the
# single-process shortcuts - every broadcast / all-gather of the multi-rank path runs, through the group's backend, with the real
# tensors.  This is how RCCL sees these collectives on a one-GPU box (it refuses two ranks on one device).
REHEARSE_ONE_RANK = False


def _collective(dist, world):
    """Does the multi-rank path (with its collectives) apply?"""
    return dist is not None and (world > 1 or REHEARSE_ONE_RANK)


class ShardedSolver:
    """Mesh rows of ONE pair sharded over the ranks of ``dist`` (None = single process).

    ``broadcast_inputs()`` = the once-per-pair exchange: the keypoint table and the
    de-normalisation block go from rank 0 to everyone (``solve()`` does it on its first call if
    the caller has not).  ``solve()`` = local solve of this rank's rows + all-gather of the H
    grid; afterwards ``self.H`` holds the full grid on every rank (what a following sharded
    warp needs for its own rows, and what rank 0 writes).

    With ``overlap=True`` the rank's rows are solved in TWO launches and the first half's all-gather runs while
    the second half is being computed.  What that buys depends on what bounds the collective: the second gather is
    exposed either way, so a latency-bound gather gains nothing and the split costs its launch overhead - measured on
    one GPU playing a rank of C4 (``tools/scaling_model.py``; round 3: tools/shard_solve_cost.py at git tag r05-hooks): +11 us at 2 ranks (715 -> 726 us), +15 at 4,
    +24 at 8 (189 -> 213 us).  The H grid is 36 B per cell: at 2 ranks a rank receives 2.9 MB over one xGMI link
    (bandwidth-bound, ~60 us: hiding half of it pays), at 8 ranks 720 KB over each of 7 links (~15 us of wire time
    under ~20 us of latency: it does not).  ``overlap="auto"`` (default) therefore splits for world sizes up to 4.
    No multi-GPU hardware was available to this builder: the rule is a model, and ``bench.py`` times both forms on
    whatever it runs on.  What hides the collective at any size is ``step()``: the gather runs beside the rank's own
    warp band.  Every buffer of the step (shard grids, gather buffers, scratch, status) is allocated here, once.

    ``same_bits=True`` makes the shards sum every cell's keypoints in the order the whole mesh would on one GPU
    (``APAP_OPT_PLAN_CELLS``): the gathered grid then equals the single-GPU grid bit for bit for any number of
    ranks.  Off by default: a shard is a small launch, and the finer keypoint splits it would otherwise get are
    what fills the GPU - the grids of the two settings differ by float64 summation order only (a float32 value in
    a few thousand may round the other way; the 1e-4 px parity bar is six orders of magnitude above that).
    """

    def __init__(self, pair, dev, dist=None, solve_fn=hip_solve, warp_fn=hip_warp_rows, ctx=None, overlap="auto",
                 same_bits=False, resident_warp=False):
        self.pair, self.dev, self.dist, self.solve_fn, self.warp_fn = pair, dev, dist, solve_fn, warp_fn
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        self.rows, self.cols = pair.vertices.shape[:2]
        self.cells_total = self.rows * self.cols
        if same_bits and solve_fn is hip_solve:
            # a context of the solver's own (the caller's keeps its options: APAP_OPT_PLAN_CELLS would change the kernel
            # choice and the keypoint splits of every later solve made with it), carrying the caller's other options over
            own = _native.Context()
            if ctx is not None:
                for name in _native.Context._NAMES:
                    own.set(name, ctx.get(name))
            own.set("plan_cells", self.cells_total)
            ctx = own
        # a _native.Context (options, profiling) handed to the default HIP compute functions
        self._ctx = ctx
        self._kw = {"ctx": ctx} if ctx is not None else {}
        self.overlap = (2 <= self.world <= 4) if overlap == "auto" else (bool(overlap) and _collective(dist, self.world))
        self.n = len(pair.src)
        self.parts = row_partition(self.rows, self.world)
        self.max_rows = max(b - a for a, b in self.parts)
        a, b = self.parts[self.rank]
        self.my_rows = (a, b)
        # the two launches of a rank: rows [a, m) and [m, b)
        halves = [(ra, ra + (rb - ra + 1) // 2, rb) for ra, rb in self.parts] if self.overlap else [(ra, rb, rb) for ra, rb in self.parts]
        self._pieces = [[(ra, rm) for ra, rm, rb in halves], [(rm, rb) for ra, rm, rb in halves]]
        verts = np.ascontiguousarray(pair.vertices.reshape(-1, 2))
        self._vert, self._mine, self._gather, self._dst, self._src = [], [], [], [], []
        for piece in self._pieces:
            pa, pb = piece[self.rank]
            pad = max(qb - qa for qa, qb in piece) * self.cols          # cells per rank in this piece, padded to the largest
            self._vert.append(torch.from_numpy(verts[pa * self.cols:pb * self.cols]).to(dev))
            self._mine.append(torch.zeros((max(pad, 1), 9), dtype=torch.float32, device=dev))
            self._gather.append(torch.zeros((self.world, max(pad, 1), 9), dtype=torch.float32, device=dev))
            # where the gathered rows go: rank r's valid cells -> its rows of the grid, the padding is dropped
            dst = np.concatenate([np.arange(qa * self.cols, qb * self.cols) for qa, qb in piece]) if self.cells_total else np.zeros(0, int)
            src = np.concatenate([r * max(pad, 1) + np.arange((qb - qa) * self.cols) for r, (qa, qb) in enumerate(piece)])
            self._dst.append(torch.from_numpy(dst.astype(np.int64)).to(dev))
            self._src.append(torch.from_numpy(src.astype(np.int64)).to(dev))
        self.vert = torch.from_numpy(verts[a * self.cols:b * self.cols]).to(dev)
        # rank 0 owns the host set-up; the others receive the result
        self.table = torch.zeros((self.n, _native.TABLE_STRIDE), dtype=torch.float64, device=dev)
        self.denorm = torch.zeros(_native.DENORM_DOUBLES, dtype=torch.float64, device=dev)
        if self.rank == 0:
            q = _native.host_prepare(pair.src, pair.dst)
            self.table.copy_(torch.from_numpy(_native.host_build_table(pair.src, q["cf1"], q["cf2"],
                                                                       moments=ctx.get("moments") if ctx is not None else 30)))
            self.denorm.copy_(torch.from_numpy(_native.host_build_denorm(q["iC2"], q["C1"], q["iN2"], q["N1"])))
        self.H = torch.zeros((self.cells_total, 9), dtype=torch.float32, device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self.cells = self.cells_total
        self.my_cells = (b - a) * self.cols          # what this rank's kernels work on
        self._inputs_sent = False
        self._pending = []
        self._solve_kw = dict(self._kw)
        if solve_fn is hip_solve:                   # the engine's scratch: sized once for the larger launch
            lib = _native.lib()
            nb = max([max(lib.apap_solve_workspace_bytes(_native._h(ctx), self.n, v.shape[0]),
                          lib.apap_solve_batch_workspace_bytes(_native._h(ctx), self.n, v.shape[0], 1), 256) if v.shape[0] else 256
                      for v in self._vert])
            self._solve_kw["work"] = torch.empty(nb, dtype=torch.uint8, device=dev)
        # the resident warp form (default engine, bands aligned to the rank's mesh rows): a WarpPlan over the rank's OWN mesh
        # rows - canvas row / column tables built once, the per-cell half in the tail of the rank's solve -, so that a warp
        # step is the gather kernel alone, as on one GPU (_make_plan)
        # The plan is made when a warp is first asked for (warp() / step(), or resident_warp=True): a caller that only wants the
        # H grid pays neither the canvas-sized workspace nor the ~160 bytes per cell the warp-ready tail of K2 writes.
        self._plan = None
        self._plan_tried = False
        self._cells_ready = False
        self._want_plan = bool(resident_warp)

    def status_word(self):
        """The device status bits of this solver's warps (the plan's geometry bits included; synchronises)."""
        geo = self._plan.geo_status if self._plan is not None else 0
        return int(self.status.cpu()[0]) | geo

    def broadcast_inputs(self):
        """Keypoint table (n x 256 B) and de-normalisation block from rank 0 to every rank:
        once per pair, not once per solve."""
        d = self.dist
        if _collective(d, self.world):
            d.broadcast(self.table, src=0)
            d.broadcast(self.denorm, src=0)
        self._inputs_sent = True

    def _band_geometry(self):
        """Canvas rows per rank.  ALIGNED TO THE RANK'S OWN MESH ROWS when the row edges allow it: band r = the
        canvas rows whose cell row rank r solved, so a rank warps from its own rows of the H grid - no wait for the all-gather -
        and its per-cell set-up covers 1 / world of the cells.  The engine sees the rank's rows as a mesh of their own: edges
        ``mesh_h[a : b + 1]`` with the last one opened to +inf on a multi-rank group (canvas rows outside the band must still
        look up SOME cell; they are not warped).  Needs increasing row edges that start at or below 0 and reach the canvas's
        last row (edges that stop short keep the unaligned bands: there the single-GPU path and the reference report an index
        error for the uncovered rows, apap.py:207 - opening the last edge would silently warp them); any other mesh keeps
        near-equal bands warped from the whole gathered grid."""
        if hasattr(self, "bands"):
            return
        p = self.pair
        edges = np.asarray(p.mesh[1], dtype=np.float64)
        self._aligned = (len(edges) == self.rows + 1 and bool(np.all(np.diff(edges) > 0)) and edges[0] <= 0.0
                         and bool(np.isfinite(edges).all()) and edges[-1] >= p.final_h)
        if self._aligned:
            first = lambda k: int(min(max(np.ceil(edges[k]), 0.0), p.final_h)) if k < self.rows else p.final_h   # noqa: E731
            self.bands = [(first(ra), first(rb)) if rb > ra else (first(ra), first(ra)) for ra, rb in self.parts]
            self.bands[0] = (0, self.bands[0][1])
        else:
            self.bands = row_partition(p.final_h, self.world)
        self.max_band = max(b - a for a, b in self.bands)
        ra, rb = self.my_rows
        if self._aligned and rb > ra:
            own = edges[ra:rb + 1].copy()
            if self.world > 1:
                own[-1] = np.inf
            self._own_edges, self._warp_shape = own, (rb - ra, self.cols)
        else:
            self._own_edges, self._warp_shape = np.ascontiguousarray(edges), (self.rows, self.cols)

    def _make_plan(self):
        """The rank's WarpPlan (None when the injected compute functions, an unaligned mesh or an empty shard rule it out)."""
        if self._plan_tried:
            return self._plan
        self._plan_tried = True
        self._band_geometry()
        ra, rb = self.my_rows
        if self.solve_fn is hip_solve and self.warp_fn is hip_warp_rows and self._aligned and rb > ra and len(self.pair.mesh[0]) <= 4096 \
                and len(self._own_edges) <= 4096:
            p = self.pair
            self._plan = WarpPlan((p.mesh[0], self._own_edges), self._warp_shape, p.final_w, p.final_h, p.off_x, p.off_y, self.dev,
                                  ctx=self._ctx)
        return self._plan

    def _solve_piece(self, k):
        """Launch piece k of this rank's rows into its (padded) shard buffer."""
        vert, mine = self._vert[k], self._mine[k]
        plan = self._make_plan() if self._want_plan and not self.overlap and self.solve_fn is hip_solve else None
        if plan is not None and vert.shape[0] == plan.rows * plan.cols:
            # the rank's rows in ONE launch whose tail leaves every cell warp ready in the plan's workspace
            plan.solve(self.table, self.denorm, vert, self.pair.gamma, self.pair.sigma, out=mine, work=self._solve_kw.get("work"))
            self._cells_ready = True
            return mine
        self._cells_ready = False
        kw = dict(self._solve_kw)
        if self.solve_fn is hip_solve:
            kw["out"] = mine
        res = self.solve_fn(self.table, self.denorm, vert, self.pair.gamma, self.pair.sigma, **kw)
        if res.data_ptr() != mine.data_ptr() and vert.shape[0]:
            mine[:vert.shape[0]].copy_(res)
        return mine

    def solve(self, stream=None, wait=True):
        """Solve this rank's rows and all-gather the H grid.  The rank's own rows are in ``self.H`` as soon as
        their kernels have run (what its own warp band needs); with ``wait=False`` the all-gathers are left in
        flight - ``finish()`` waits for them and puts the other ranks' rows in place."""
        d = self.dist
        if not self._inputs_sent:
            self.broadcast_inputs()
        self.finish()
        if not _collective(d, self.world):
            mine = self._solve_piece(0)
            self.H.copy_(mine[:self.cells_total])
            return self.H
        for k in range(2 if self.overlap else 1):
            mine = self._solve_piece(k)
            pa, pb = self._pieces[k][self.rank]
            if pb > pa:
                self.H[pa * self.cols:pb * self.cols].copy_(mine[:(pb - pa) * self.cols])
            # async: the collective is ordered after this piece's kernels and runs beside whatever comes next
            self._pending.append((k, d.all_gather_into_tensor(self._gather[k].view(-1, 9), mine, async_op=True)))
        if wait:
            self.finish()
        return self.H

    def finish(self):
        """Wait for the all-gathers ``solve(wait=False)`` left in flight; afterwards ``self.H`` is the whole grid."""
        for k, w in self._pending:
            w.wait()
            g = self._gather[k].view(-1, 9)
            if self._src[k].numel() != g.shape[0]:      # uneven shards: drop the padding
                g = g.index_select(0, self._src[k])
            self.H.index_copy_(0, self._dst[k], g)
        self._pending = []
        return self.H

    def step(self, gather_canvas=False):
        """One pipelined step of the pair: solve this rank's rows, start the all-gather of the H grid, warp this
        rank's band FROM ITS OWN ROWS while the gather runs, then wait for the gather.  Returns ``(H, band or canvas)``."""
        self._want_plan = True
        self.solve(wait=False)
        band = self.warp(gather=gather_canvas)
        return self.finish(), band

    def _warp_setup(self):
        """Once per pair: the source image reaches every rank by a broadcast from rank 0 (25 MB at 4K, 100 MB at 8K
        - the transfer SURVEY.md 8e warns dominates a single warp); bands, buffers and the rank's warp plan."""
        p, d = self.pair, self.dist
        self._band_geometry()
        if self.rank == 0 and p.img is not None:
            self.img = torch.from_numpy(np.ascontiguousarray(p.img)).to(self.dev)
        else:
            self.img = torch.zeros(p.shape, dtype=torch.uint8, device=self.dev)
        if _collective(d, self.world):
            d.broadcast(self.img, src=0)
        self.mesh_w = torch.from_numpy(np.ascontiguousarray(p.mesh[0])).to(self.dev)
        self.mesh_h = torch.from_numpy(self._own_edges).to(self.dev)
        if self.warp_fn is hip_warp_rows:       # the engine's scratch and status word: once
            if self._make_plan() is None:       # (the plan carries its own)
                nb = _native.lib().apap_warp_workspace_bytes(self._warp_shape[0], self._warp_shape[1], p.final_w, p.final_h)
                self._warp_kw = dict(self._kw, work=torch.empty(nb, dtype=torch.uint8, device=self.dev), status=self.status)
        else:
            self._warp_kw = dict(self._kw)
        self._band = torch.zeros((max(self.max_band, 1), p.final_w, 3), dtype=torch.uint8, device=self.dev)
        self._bands = torch.zeros((self.world, max(self.max_band, 1), p.final_w, 3), dtype=torch.uint8, device=self.dev)
        dst = np.concatenate([np.arange(ba, bb) for ba, bb in self.bands])
        src = np.concatenate([r * max(self.max_band, 1) + np.arange(bb - ba) for r, (ba, bb) in enumerate(self.bands)])
        self._band_dst = torch.from_numpy(dst.astype(np.int64)).to(self.dev)
        self._band_src = torch.from_numpy(src.astype(np.int64)).to(self.dev)
        self.out = torch.zeros((p.final_h, p.final_w, 3), dtype=torch.uint8, device=self.dev)

    def warp(self, stream=None, gather=True):
        """Backward warp of the pair with canvas rows sharded over the ranks: every rank warps its band - from its
        own rows of the H grid when the bands are aligned to the mesh rows (``_warp_setup``), from the whole
        gathered grid otherwise; one all-gather assembles the canvas on every rank.  Returns ``self.out``.

        ``gather=False`` stops after the band: the canvas stays distributed (rank r holds rows
        ``self.bands[r]`` in ``self._band``) - what a pipeline that writes or consumes the bands in
        place does, and the part of the step that scales; returns this rank's band."""
        self._want_plan = True          # from now on the solve's tail leaves the rank's cells warp ready
        if not hasattr(self, "img"):
            self._warp_setup()
        p, d = self.pair, self.dist
        a, b = self.bands[self.rank]
        single = not _collective(d, self.world)
        if self._aligned:
            ra, rb = self.my_rows
            H = self.H[ra * self.cols:rb * self.cols]           # this rank's own rows: valid before the gather ends
        else:
            self.finish()
            H = self.H
        if (b > a or single) and self._plan is not None:
            # resident form: the per-cell tables are in the plan's workspace (left there by the solve's tail, or built here
            # from the rank's rows of the grid when the solve ran in two pieces) - the step is the gather kernel alone
            if not self._cells_ready:
                self._plan.cells(H)
                self._cells_ready = True
            target = self.out if single else self._band[:b - a]       # one rank: straight into the canvas
            self._plan.gather(self.img, out=target.view(1, b - a, p.final_w, 3), rows=(a, b - a))
            self.status = self._plan.status
        elif b > a or single:
            st = self.warp_fn(self.img, H, self.mesh_w, self.mesh_h, p.final_w, p.final_h, p.off_x, p.off_y, a, b - a,
                              self.out if single else self._band, self._warp_shape, **self._warp_kw)   # one rank: straight into the canvas
            if st is not None:
                self.status = st
        if single:
            return self.out
        if not gather:
            return self._band[:b - a]
        if all(bb - ba == b - a for ba, bb in self.bands):
            d.all_gather_into_tensor(self.out, self._band[:b - a])
            return self.out
        d.all_gather_into_tensor(self._bands.view(-1, p.final_w, 3), self._band)
        self.out.index_copy_(0, self._band_dst, self._bands.view(-1, p.final_w, 3).index_select(0, self._band_src))   # drop the padding
        return self.out


def solve_pairs(pairs, dev, dist=None, solve_fn=hip_solve, ctx=None):
    """Independent pairs dealt round-robin to the ranks; returns, on rank 0, the list of
    H grids in input order (``None`` elsewhere).  All pairs must share one mesh shape.  ``ctx``: the solver options (a
    context with ``moments=24`` gets the 24-sum tables it needs)."""
    moments = ctx.get("moments") if ctx is not None else 30
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    mine, tables, dens = [], [], []
    my_pairs = [pairs[k] for k in range(rank, len(pairs), world)]
    for p in my_pairs:
        q = _native.host_prepare(p.src, p.dst)
        tables.append(torch.from_numpy(_native.host_build_table(p.src, q["cf1"], q["cf2"], moments=moments)))
        dens.append(torch.from_numpy(_native.host_build_denorm(q["iC2"], q["C1"], q["iN2"], q["N1"])))
    same = (len(my_pairs) > 1 and solve_fn is hip_solve
            and all(len(p.src) == len(my_pairs[0].src) and np.array_equal(p.vertices, my_pairs[0].vertices)
                    and (p.gamma, p.sigma) == (my_pairs[0].gamma, my_pairs[0].sigma) for p in my_pairs))
    if same:   # one batched launch for this rank's pairs
        vert = torch.from_numpy(np.ascontiguousarray(my_pairs[0].vertices.reshape(-1, 2))).to(dev)
        Hb = hip_solve_batch(torch.stack(tables).to(dev), torch.stack(dens).to(dev), vert, my_pairs[0].gamma,
                             my_pairs[0].sigma, ctx=ctx)
        mine = list(Hb)
    else:
        for p, table, den in zip(my_pairs, tables, dens):
            vert = torch.from_numpy(np.ascontiguousarray(p.vertices.reshape(-1, 2))).to(dev)
            mine.append(solve_fn(table.to(dev), den.to(dev), vert, p.gamma, p.sigma, **({"ctx": ctx} if ctx is not None else {})))
    if not _collective(dist, world):
        return [h.cpu().numpy().reshape(p.vertices.shape[0], p.vertices.shape[1], 3, 3) for h, p in zip(mine, pairs)]
    per_rank = (len(pairs) + world - 1) // world
    cells = pairs[0].vertices.shape[0] * pairs[0].vertices.shape[1]
    buf = torch.zeros((per_rank, cells, 9), dtype=torch.float32, device=dev)
    for i, h in enumerate(mine):
        buf[i].copy_(h)
    out = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
    if dist.get_backend() == "nccl":
        allbuf = torch.zeros((world,) + tuple(buf.shape), dtype=buf.dtype, device=dev)
        dist.all_gather_into_tensor(allbuf.view(-1, cells, 9), buf)
        out = list(allbuf) if rank == 0 else None
    else:
        dist.gather(buf, out, dst=0)
    if rank != 0:
        return None
    rows, cols = pairs[0].vertices.shape[:2]
    return [out[k % world][k // world].cpu().numpy().reshape(rows, cols, 3, 3) for k in range(len(pairs))]


def warp_pairs(pairs, grids, dev, dist=None, warp_fn=hip_warp_batch, gather=False, ctx=None, plans=None):
    """The warp half of independent pairs (BASELINE config 5; the reference runs apap.py:186-217 once per pair): the
    pairs are dealt round-robin to the ranks like ``solve_pairs`` deals them, every rank warps ITS pairs in one batched
    set of launches (grid.z = pair), no collective on the data path.  ``grids[k]`` is pair k's H grid (rows, cols, 3, 3)
    - only the entries of this rank's pairs are read (None elsewhere is fine).  All pairs must share the image size, the
    mesh and the canvas geometry.  Returns ``{pair index: canvas (final_h, final_w, 3) uint8 tensor on dev}`` for this
    rank's pairs - 27 MB per 4K canvas: they stay where they were computed - or, with ``gather=True``, on rank 0 the
    list of all canvases as numpy arrays in input order (``None`` on the other ranks): for tests and small batches.
    ``plans``: a dict the CALLER keeps between calls - the rank's WarpPlan (workspace + the canvas row / column tables of this
    geometry) is left in it, so that a later call on the same geometry, share of pairs and context skips the geometry phase and
    allocates no workspace (default engine only; the stacks of grids and images and the canvases are still made per call)."""
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    mine = list(range(rank, len(pairs), world))
    p0 = pairs[0]
    for p in pairs:
        if (p.shape, p.final_w, p.final_h, p.off_x, p.off_y) != (p0.shape, p0.final_w, p0.final_h, p0.off_x, p0.off_y) \
                or not np.array_equal(p.mesh, p0.mesh):
            raise ValueError("warp_pairs: the pairs of a batch must share image size, mesh edges, canvas size and offsets")
    rows, cols = p0.vertices.shape[:2]
    canv = {}
    if mine:
        imgs = torch.stack([torch.from_numpy(np.ascontiguousarray(pairs[k].img)) for k in mine]).to(dev)
        H = torch.stack([torch.from_numpy(np.ascontiguousarray(grids[k], dtype=np.float32).reshape(rows * cols, 9)) for k in mine]).to(dev)
        mesh_w = torch.from_numpy(np.ascontiguousarray(p0.mesh[0], dtype=np.float64)).to(dev)
        mesh_h = torch.from_numpy(np.ascontiguousarray(p0.mesh[1], dtype=np.float64)).to(dev)
        kw = {"ctx": ctx} if ctx is not None else {}
        if plans is not None and warp_fn is hip_warp_batch and len(p0.mesh[0]) <= 4096 and len(p0.mesh[1]) <= 4096:
            # (a plan is bound to its context's options and to the dtype the edges were given in)
            key = (rows, cols, p0.final_w, p0.final_h, p0.off_x, p0.off_y, len(mine), str(dev), id(ctx), p0.mesh.dtype.str,
                   p0.mesh.tobytes())
            plan = plans.get(key)
            if plan is None:
                plan = plans[key] = WarpPlan(p0.mesh, (rows, cols), p0.final_w, p0.final_h, p0.off_x, p0.off_y, dev, batch=len(mine), ctx=ctx)
            plan.begin()
            plan.cells(H.view(-1, 9))
            out, word = plan.gather(imgs), plan.status_word()
        else:
            out, status = warp_fn(imgs, H, mesh_w, mesh_h, p0.final_w, p0.final_h, p0.off_x, p0.off_y, (rows, cols), **kw)
            word = int(status.cpu()[0]) if status is not None else 0
        if word != 0:
            code = _native.ERR_SINGULAR if word & 1 else _native.ERR_INDEX if word & 2 else _native.ERR_INVALID_ARG
            raise _native._ERROR_CLASSES[code](code, "warp_pairs: device status word %d" % word)
        canv = {k: out[i] for i, k in enumerate(mine)}
    if not gather:
        return canv
    per_rank = (len(pairs) + world - 1) // world
    buf = torch.zeros((per_rank, p0.final_h, p0.final_w, 3), dtype=torch.uint8, device=dev)
    for i, k in enumerate(mine):
        buf[i].copy_(canv[k])
    if not _collective(dist, world):
        return [buf[k].cpu().numpy() for k in range(len(pairs))]
    if dist.get_backend() == "nccl":
        allbuf = torch.zeros((world,) + tuple(buf.shape), dtype=buf.dtype, device=dev)
        dist.all_gather_into_tensor(allbuf.view((-1,) + tuple(buf.shape[1:])), buf)
        got = list(allbuf) if rank == 0 else None
    else:
        got = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
        dist.gather(buf, got, dst=0)
    if rank != 0:
        return None
    return [got[k % world][k // world].cpu().numpy() for k in range(len(pairs))]
