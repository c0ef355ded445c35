"""CPU check of the claim the default warp kernel's exactness rests on (oracle/warp_fast_spec.py):
a pixel that the float32 estimate does NOT flag "in doubt" has exactly the reference's integer source
pixel, whatever legal value v_rcp_f32 returns.  Flagged pixels are recomputed with the exact float64
sequence on the device, so only the unflagged ones need this proof; the GPU tests then compare whole
canvases with the reference's (SHA-256)."""
import numpy as np
import pytest

from oracle import warp_fast_spec as S
from cvx_proj_amd.synth import config_pair


def check_cells(h, xb, yb, DX, DY, rng, samples=64):
    """Records for the cells `h` (n, 9), then `samples` pixels per cell inside the cell's extent."""
    n = len(h)
    rec = S.record(h, np.ones(n, bool), xb, yb, DX, DY)
    worst, doubts, total = 0.0, 0, 0
    for ulps in (-1, 0, 1, None):
        dx = rng.integers(-DX.astype(int), DX.astype(int) + 1, size=(samples, n))
        dy = rng.integers(-DY.astype(int), DY.astype(int) + 1, size=(samples, n))
        r = rng.integers(-1, 2, size=(samples, n)) if ulps is None else ulps
        ix, iy, doubt = S.estimate(rec, dx, dy, r)
        tx, ty = S.reference_coords(h, xb + dx, yb + dy)
        sure = ~doubt
        assert not (sure & ~rec["good"]).any(), "a cell without a bound must flag every pixel"
        with np.errstate(invalid="ignore"):
            assert (np.floor(tx[sure]) == ix[sure]).all() and (np.floor(ty[sure]) == iy[sure]).all()
            assert (tx[sure] != np.floor(tx[sure])).all() and (ty[sure] != np.floor(ty[sure])).all()
        doubts += int(doubt[:, rec["good"]].sum())
        total += int(rec["good"].sum()) * samples
    return rec, doubts / max(total, 1)


@pytest.mark.parametrize("cfg", ["C1", "C2"])
def test_every_unflagged_pixel_of_a_baseline_canvas(golden, cfg):
    """Every pixel of the C1 / C2 canvas through the restated record + float32 arithmetic."""
    g = golden(cfg.lower() + "_ref")
    p = config_pair(cfg, with_image=False)
    H = g["H_ref"]
    hinv = np.linalg.inv(H.astype(np.float64)).astype(np.float32).astype(np.float64).reshape(H.shape[0], H.shape[1], 9)
    okx, x0, sx = S.origin(p.mesh[0], p.final_w)
    oky, y0, sy = S.origin(p.mesh[1], p.final_h)
    assert okx.all() and oky.all()
    rec = S.record(hinv, np.ones(H.shape[:2], bool), (x0 + sx // 2 - p.off_x)[None, :].astype(float),
                   (y0 + sy // 2 - p.off_y)[:, None].astype(float), (sx - sx // 2)[None, :].astype(float),
                   (sy - sy // 2)[:, None].astype(float))
    assert rec["good"].all()
    # pixel -> cell, like the set-up kernel's tables
    jj, ii = np.arange(p.final_w), np.arange(p.final_h)
    cc = np.searchsorted(p.mesh[0], jj, side="right") - 1
    cr = np.searchsorted(p.mesh[1], ii, side="right") - 1
    dx = (jj - (x0 + sx // 2)[cc])[None, :]
    dy = (ii - (y0 + sy // 2)[cr])[:, None]
    assert (np.abs(dx) <= (sx - sx // 2)[cc][None, :]).all() and (np.abs(dy) <= (sy - sy // 2)[cr][:, None]).all()
    cell = {k: v[cr[:, None], cc[None, :]] for k, v in rec.items()}
    rng = np.random.default_rng(5)
    frac = []
    for ulps in (-1, 0, 1, rng.integers(-1, 2, size=(p.final_h, p.final_w))):
        ix, iy, doubt = S.estimate(cell, dx, dy, ulps)
        tx, ty = S.reference_coords(hinv[cr[:, None], cc[None, :]], (jj - p.off_x)[None, :].astype(float),
                                    (ii - p.off_y)[:, None].astype(float))
        sure = ~doubt
        assert (np.floor(tx[sure]) == ix[sure]).all() and (np.floor(ty[sure]) == iy[sure]).all()
        assert (tx[sure] != ix[sure]).all() and (ty[sure] != iy[sure]).all()
        frac.append(doubt.mean())
    print(f"{cfg}: {max(frac):.2e} of the pixels in doubt (exact path), window {2 * rec['du'].max() / S.UNIT:.2e} px")
    assert max(frac) < 5e-4


def test_random_projective_cells():
    """Rotations, anisotropic scales, perspective up to the limit the record accepts, anchors far from the origin."""
    rng = np.random.default_rng(11)
    n = 20000
    ang = rng.uniform(-np.pi, np.pi, n)
    s1, s2 = np.exp(rng.uniform(-2, 2, n)), np.exp(rng.uniform(-2, 2, n))
    h = np.empty((n, 9))
    h[:, 0], h[:, 1] = s1 * np.cos(ang), -s2 * np.sin(ang)
    h[:, 3], h[:, 4] = s1 * np.sin(ang), s2 * np.cos(ang)
    h[:, 2], h[:, 5] = rng.uniform(-4000, 4000, n), rng.uniform(-4000, 4000, n)
    h[:, 6], h[:, 7] = rng.normal(0, 1, n) * 10.0 ** rng.uniform(-7, -2.5, n), rng.normal(0, 1, n) * 10.0 ** rng.uniform(-7, -2.5, n)
    h[:, 8] = rng.uniform(0.5, 2.0, n) * rng.choice([-1.0, 1.0], n)
    h = h.astype(np.float32).astype(np.float64)      # the stored inverses are float32 values
    xb, yb = np.round(rng.uniform(-500, 8000, n)), np.round(rng.uniform(-500, 5000, n))
    DX, DY = rng.integers(1, 128, n).astype(float), rng.integers(1, 128, n).astype(float)
    rec, frac = check_cells(h, xb, yb, DX, DY, rng)
    print(f"random cells: {rec['good'].mean():.3f} with a bound, {frac:.2e} of their pixels in doubt")
    assert rec["good"].mean() > 0.5


def test_estimates_near_the_fixed_point_limit():
    """The 16.16 estimate keeps a SIGNED 16-bit integer half: the record must refuse a cell whose estimate could leave it
    (|estimate| bound >= 500 px, kFastMaxEstimate of apap_kernels.hip), and be exact right up to that cap - magnifications of
    2 ... 6 over cells of up to 127 px put the bound between ~250 and ~800 px on either side of it."""
    rng = np.random.default_rng(23)
    n = 6000
    mag = rng.uniform(2.0, 6.5, n)
    ang = rng.uniform(-np.pi, np.pi, n)
    h = np.zeros((n, 9))
    h[:, 0], h[:, 1] = mag * np.cos(ang), -mag * np.sin(ang)
    h[:, 3], h[:, 4] = mag * np.sin(ang), mag * np.cos(ang)
    h[:, 2], h[:, 5] = rng.uniform(-2000, 2000, n), rng.uniform(-2000, 2000, n)
    h[:, 6], h[:, 7] = rng.normal(0, 1e-5, n), rng.normal(0, 1e-5, n)
    h[:, 8] = 1.0
    h = h.astype(np.float32).astype(np.float64)
    xb, yb = np.round(rng.uniform(0, 4000, n)), np.round(rng.uniform(0, 2000, n))
    DX, DY = rng.integers(90, 128, n).astype(float), rng.integers(90, 128, n).astype(float)
    rec, frac = check_cells(h, xb, yb, DX, DY, rng, samples=48)
    reach = mag * np.hypot(DX, DY)                    # how far from the anchor a pixel of the cell can land
    assert rec["good"][reach < 330].all(), "cells well inside the cap must keep their bound"
    assert not rec["good"][mag * np.maximum(DX, DY) > 520].any(), "cells beyond the cap must take the exact path"
    assert 0.2 < rec["good"].mean() < 0.9
    print(f"near the limit: {rec['good'].mean():.3f} of the cells keep a bound, {frac:.2e} of their pixels in doubt")


def test_cells_without_a_bound_flag_everything():
    """Denominator through zero inside the cell, non-finite entries, coordinates beyond 2^30, zero matrix."""
    rng = np.random.default_rng(2)
    h = np.array([[1, 0, 0, 0, 1, 0, 0.02, 0, -1.0],          # t2 = 0.02 x - 1 changes sign near x = 50
                  [1, 0, 0, 0, 1, 0, 0, 0, 0],                  # t2 = 0
                  [np.nan, 0, 0, 0, 1, 0, 0, 0, 1],
                  [1e12, 0, 0, 0, 1, 0, 0, 0, 1],               # coordinates beyond 2^30
                  [0, 0, 0, 0, 0, 0, 0, 0, 0],
                  [1, 0, 0, 0, 1, 0, 4e-3, 0, 1.0]], float)     # t2 moves by more than a quarter inside the cell: refused
    xb, yb = np.full(6, 50.0), np.full(6, 10.0)
    DX, DY = np.array([20.0, 5, 5, 5, 5, 127]), np.full(6, 5.0)
    rec, _ = check_cells(h, xb, yb, DX, DY, rng, samples=16)
    assert not rec["good"][:5].any() and (rec["thr"][:5] == 0xffffffff).all()
    assert not rec["good"][5]


def test_origin_rejects_irregular_edges():
    ok, x0, span = S.origin(np.array([0.0, 10.5, 10.5, 8.0, np.nan, 30.0, 400.0]), 300)
    assert ok.tolist() == [True, False, False, False, False, True]
    assert x0[0] == 0 and span[0] == 11 and x0[5] == 30 and span[5] == 254      # clamped to a byte
    ok, x0, span = S.origin(np.array([-0.5, 3.0, 1e12]), 50)
    assert ok.tolist() == [True, True] and x0.tolist() == [0, 3] and span.tolist() == [3, 47]
