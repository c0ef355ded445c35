"""K3's persistent column-walk form (``APAP_OPT_WARP_WALK = 1``: a grid sized to the chip, every wave walks down a contiguous
share of the canvas rows of one 256-pixel column block, software-pipelined) against the reference's canvases, the oracle and
the strip kernels - byte for byte, for every stage depth, wave count and run length, on regular and irregular meshes, bands,
batches and the fused stitch.  Run on the GPU box: ``python -m pytest tests -m gpu``."""
import hashlib

import numpy as np
import pytest

from oracle import apap_oracle as O
from cvx_proj_amd.synth import config_pair, synth_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def need_gpu(native):
    assert native.lib().apap_device_count() >= 1, "these tests need a GPU; the library found none"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest()


FORMS = [dict(warp_stage=2), dict(warp_stage=1), dict(warp_stage=2, warp_waves=8), dict(warp_stage=2, warp_waves=32, warp_min_run=1),
         dict(warp_stage=1, warp_min_run=3), dict(warp_stage=1, warp_min_run=100000)]


@pytest.fixture(params=FORMS, ids=lambda d: ",".join(f"{k[5:]}={v}" for k, v in d.items()))
def walk(request, native):
    ctx = native.Context(warp_walk=1, **request.param)
    yield ctx
    ctx.close()


@pytest.mark.parametrize("name", ["tiny_sigma100", "tiny_sigma6"])
def test_tiny_warp_vs_reference(native, golden, walk, name):
    g = golden(name)
    fw, fh, ox, oy = (int(v) for v in g["final"])
    warped, hinv = native.local_warp(g["img"], g["H_ref"], g["mesh"][0], g["mesh"][1], fw, fh, ox, oy, ctx=walk)
    assert np.array_equal(hinv, g["Hinv_ref"])
    assert np.array_equal(warped, g["warped_ref"])


def test_warp_edge_cases_vs_reference(native, golden, walk):
    g = golden("warp_edge_ref")
    for k in range(int(g["count"])):
        fw, fh, ox, oy = (int(v) for v in g[f"geo{k}"])
        out, hinv = native.local_warp(g[f"img{k}"], g[f"H{k}"].copy(), g[f"mesh_w{k}"], g[f"mesh_h{k}"], fw, fh, ox, oy, ctx=walk)
        assert np.array_equal(hinv, g[f"Hinv{k}"]), k
        assert np.array_equal(out, g[f"warped{k}"]), k


@pytest.mark.parametrize("cfg,name", [("C1", "c1_ref"), ("C2", "c2_ref"), ("C3", "c3_ref")])
def test_config_canvases_vs_reference(native, golden, walk, cfg, name):
    """The canvases of BASELINE's configurations: SHA-256 and sampled rows of the reference's own pixel loop."""
    g = golden(name)
    p = config_pair(cfg)
    w, hinv = native.local_warp(p.img, g["H_ref"], p.mesh[0], p.mesh[1], p.final_w, p.final_h, p.off_x, p.off_y, ctx=walk)
    every = int(g["warp_rows_every"])
    assert np.array_equal(w[::every], g["warped_rows"])
    assert sha(w) == g["warped_sha256"].tobytes()
    assert sha(hinv) == g["Hinv_sha256"].tobytes() if "Hinv_sha256" in g else True


def test_c4_canvas_vs_reference(native, golden):
    """The 8018 x 4485 canvas of config 4 (36 Mpix: long runs per wave) - SHA-256 of the reference's pixel loop."""
    g = golden("c4_ref_rows8")
    p = config_pair("C4")
    H, _ = native.local_homography(p.src, p.dst, p.vertices, p.gamma, p.sigma, want_weights=False)
    assert sha(H) == g["H_sha256"].tobytes()
    ctx = native.Context(warp_walk=1)
    try:
        w, _ = native.local_warp(p.img, H, p.mesh[0], p.mesh[1], p.final_w, p.final_h, p.off_x, p.off_y, ctx=ctx)
    finally:
        ctx.close()
    assert sha(w) == g["warped_sha256"].tobytes()


@pytest.mark.parametrize("seed", range(12))
def test_irregular_meshes_and_strong_perspective(native, walk, seed):
    """Inputs the float32 estimate has no bound for (edges out of order, repeated edges, cells wider than 254 pixels or
    narrower than a lane's four, denominators that change sign, huge entries): every pixel of such a cell goes through the
    exact path inside the pipelined loop.  Walk form == all-float64 strip kernel == oracle."""
    rng = np.random.default_rng(9000 + seed)
    ih, iw = int(rng.integers(40, 300)), int(rng.integers(40, 700))
    img = rng.integers(1, 256, (ih, iw, 3), dtype=np.uint8)
    fw, fh = int(rng.integers(5, 900)), int(rng.integers(5, 260))
    rows, cols = int(rng.integers(1, 12)), int(rng.integers(1, 40))

    def edges(n, size):
        e = np.sort(rng.uniform(0, size, n - 2))
        e = np.concatenate([[0.0], e, [float(size)]])
        kind = seed % 4
        if kind == 1 and n > 2:
            e[1] = e[2]
            if n > 4:
                e[3], e[4] = e[4], e[3]
        elif kind == 2:
            e = np.round(e)
            e[-1] = size
        elif kind == 3 and n > 3:
            e[1:4] = [1.0, 2.0, 3.0]
        return e
    mesh_w, mesh_h = edges(cols + 1, fw), edges(rows + 1, fh)
    mesh_w[-1] = max(mesh_w.max(), fw)
    mesh_h[-1] = max(mesh_h.max(), fh)
    H = np.empty((rows, cols, 3, 3), np.float32)
    for r in range(rows):
        for c in range(cols):
            a = rng.uniform(-0.4, 0.4)
            s = np.exp(rng.uniform(-0.5, 0.5))
            persp = rng.normal(0, 1, 2) * 10.0 ** rng.uniform(-6, -1.5)
            H[r, c] = [[s * np.cos(a), -s * np.sin(a), rng.uniform(-30, 30)],
                       [s * np.sin(a), s * np.cos(a), rng.uniform(-30, 30)],
                       [persp[0], persp[1], 1.0]]
    if seed % 3 == 0:
        H[0, 0] *= 1e12
    ox, oy = int(rng.integers(-20, 20)), int(rng.integers(-20, 20))
    exact_ctx = native.Context(warp_fast=0)
    try:
        out_w, hinv_w = native.local_warp(img, H.copy(), mesh_w, mesh_h, fw, fh, ox, oy, ctx=walk)
        out_e, hinv_e = native.local_warp(img, H.copy(), mesh_w, mesh_h, fw, fh, ox, oy, ctx=exact_ctx)
    finally:
        exact_ctx.close()
    assert np.array_equal(hinv_w, hinv_e)
    assert np.array_equal(out_w, out_e)
    ref = O.local_warp_fast(img, hinv_w, (mesh_w, mesh_h), (fw, fh), (ox, oy))        # the oracle on the engine's own inverses
    assert np.array_equal(out_w, ref)


def test_the_image_s_last_pixel_is_patched_not_fetched(native, walk):
    """The walk form reads source pixels through a buffer descriptor that ends with the image, and the dword of the very last
    pixel would reach one byte past it: that pixel goes through the doubt path and is patched from a value read once per
    wave.  A magnifying warp whose whole lower right corner maps to the last source pixel, on images whose byte size is
    and is not a multiple of the page size."""
    for (ih, iw) in ((32, 128), (33, 77), (2, 1)):
        rng = np.random.default_rng(ih * 1000 + iw)
        img = rng.integers(1, 256, (ih, iw, 3), dtype=np.uint8)
        fw, fh = 4 * iw + 3, 4 * ih + 2
        # canvas (x, y) -> source (x / 4 + 0.3, y / 4 + 0.3): H maps source -> canvas
        H = np.tile(np.array([[4, 0, -1.2], [0, 4, -1.2], [0, 0, 1]], np.float32), (3, 2, 1, 1))
        mesh_w, mesh_h = np.linspace(0, fw, 3), np.linspace(0, fh, 4)
        out, hinv = native.local_warp(img, H, mesh_w, mesh_h, fw, fh, 0, 0, ctx=walk)
        ref = O.local_warp_fast(img, hinv, (mesh_w, mesh_h), (fw, fh), (0, 0))
        assert np.array_equal(out, ref)
        assert (out == img[-1, -1]).all(axis=-1).sum() >= 4          # the last source pixel is on the canvas, several times


def test_bands_batches_and_stitch(native, walk):
    """Row bands (what a rank of a sharded pair warps), a batch of pairs in one launch, the fused stitch: each against
    the strip kernels."""
    import torch
    from cvx_proj_amd.dist import hip_warp_batch
    dev = torch.device("cuda:0")
    pairs = [synth_pair(520, 300, 90, 11, seed=60 + k) for k in range(3)]
    p0 = pairs[0]
    rows, cols = p0.vertices.shape[:2]
    grids = [native.local_homography(p.src, p.dst, p.vertices, p.gamma, p.sigma, want_weights=False)[0] for p in pairs]
    H = torch.stack([torch.from_numpy(g.reshape(-1, 9)) for g in grids]).to(dev)
    imgs = torch.stack([torch.from_numpy(p.img) for p in pairs]).to(dev)
    mw, mh = torch.from_numpy(p0.mesh[0].copy()).to(dev), torch.from_numpy(p0.mesh[1].copy()).to(dev)
    geo = (p0.final_w, p0.final_h, p0.off_x, p0.off_y, (rows, cols))
    ref, _ = hip_warp_batch(imgs, H, mw, mh, *geo)                       # strips
    out, st = hip_warp_batch(imgs, H, mw, mh, *geo, ctx=walk)
    assert int(st.cpu()[0]) == 0 and torch.equal(out, ref)
    for (a, n) in ((0, 1), (5, 64), (p0.final_h - 3, 3), (17, p0.final_h - 17)):
        band, _ = hip_warp_batch(imgs, H, mw, mh, *geo, ctx=walk, rows=(a, n))
        assert torch.equal(band, ref[:, a:a + n]), (a, n)
    rng = np.random.default_rng(5)
    centers = rng.integers(0, 256, (3,) + p0.shape, dtype=np.uint8)
    centers[rng.random(centers.shape[:3]) < 0.2] = 0
    ct = torch.from_numpy(centers).to(dev)
    s_ref, _ = hip_warp_batch(imgs, H, mw, mh, *geo, centers=ct)
    s_out, _ = hip_warp_batch(imgs, H, mw, mh, *geo, centers=ct, ctx=walk)
    assert torch.equal(s_out, s_ref)
    k = 1
    one, _ = native.local_stitch(pairs[k].img, centers[k], grids[k], p0.mesh[0], p0.mesh[1], p0.final_w, p0.final_h, p0.off_x, p0.off_y,
                                 ctx=walk)
    assert np.array_equal(one, s_ref[k].cpu().numpy())


def test_c5_batched_canvases_vs_reference(native, golden):
    """Config 5 in the walk form: pairs 0 and 1 of a 4-pair batch against the reference's canvases."""
    import torch
    from cvx_proj_amd.dist import solve_pairs, warp_pairs
    dev = torch.device("cuda:0")
    pairs = [config_pair("C5", seed_offset=k) for k in range(4)]
    grids = solve_pairs(pairs, dev)
    ctx = native.Context(warp_walk=1)
    try:
        canv = warp_pairs(pairs, grids, dev, ctx=ctx)
    finally:
        ctx.close()
    for k in range(2):
        g = golden(f"c5_warp_k{k}")
        c = canv[k].cpu().numpy()
        assert np.array_equal(c[::int(g["warp_rows_every"])], g["warped_rows"])
        assert sha(c) == g["warped_sha256"].tobytes()
