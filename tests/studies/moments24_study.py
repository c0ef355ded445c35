"""CPU study behind the opt-in 24-sum K1 (SURVEY.md section 8a's structure note): how far the float32 grids move when
the normal matrix is built from the 24 exact-product sums instead of the 30 sums that keep apap.py:103-119's
float32-rounded products, and when the weights are evaluated in float32.  Test infrastructure (imports oracle/)."""
import sys
import numpy as np
sys.path.insert(0, ".")      # run from the repository root: python tests/studies/moments24_study.py
from oracle import apap_oracle as O
from cvx_proj_amd.synth import config_pair


def solve(p, vertices, mode, w32=False, chunk=2048):
    src, dst = p.src, p.dst
    n = src.shape[0]
    pr = O.prepare(src, dst)
    if mode == 30:
        P = O.moments_from_rows(pr["aa"]).reshape(n, 81)
    else:
        cf1 = pr["cf1"].astype(np.float64)
        cf2 = pr["cf2"].astype(np.float64)
        x, y = cf1[:, 0], cf1[:, 1]
        nxp, nyp = -cf2[:, 0], -cf2[:, 1]
        z = np.zeros(n)
        o = np.ones(n)
        r1 = np.stack([x, y, o, z, z, z, nxp * x, nxp * y, nxp], 1)
        r2 = np.stack([z, z, z, x, y, o, nyp * x, nyp * y, nyp], 1)
        P = (r1[:, :, None] * r1[:, None, :] + r2[:, :, None] * r2[:, None, :]).reshape(n, 81)
    inv = 1.0 / p.sigma ** 2
    v = vertices.reshape(-1, 2)
    s = src.astype(np.float64)
    H = np.zeros((v.shape[0], 3, 3), np.float32)
    iC2, C1, iN2, N1 = (pr[k].astype(np.float64) for k in ("iC2", "C1", "iN2", "N1"))
    for lo in range(0, v.shape[0], chunk):
        vv = v[lo:lo + chunk]
        if w32:
            dx = (vv[:, None, 0].astype(np.float32) - s[None, :, 0].astype(np.float32))
            dy = (vv[:, None, 1].astype(np.float32) - s[None, :, 1].astype(np.float32))
            w2 = np.exp(-(np.sqrt(dx * dx + dy * dy) * np.float32(2 * inv))).astype(np.float32)
            w2 = np.maximum(w2, np.float32(p.gamma ** 2)).astype(np.float64)
        else:
            dx = vv[:, None, 0] - s[None, :, 0]
            dy = vv[:, None, 1] - s[None, :, 1]
            w = np.exp(-(np.sqrt(dx ** 2 + dy ** 2) * inv))
            w[w < p.gamma] = p.gamma
            w2 = w * w
        M = (w2 @ P).reshape(-1, 9, 9)
        _, vec = np.linalg.eigh(M)
        h = vec[:, :, 0].reshape(-1, 3, 3)
        h = iC2 @ h @ C1
        h = iN2 @ h @ N1
        H[lo:lo + chunk] = h / h[:, 2:3, 2:3]
    return H.reshape(vertices.shape[:2] + (3, 3))


if __name__ == "__main__":
    for name, stride in (("C1", 1), ("C2", 3), ("C3", 8), ("C4", 20)):
        p = config_pair(name, with_image=False)
        vert = p.vertices[::stride, ::stride]
        ref = solve(p, vert, 30)
        for mode, w32 in ((24, False), (30, True), (24, True)):
            H = solve(p, vert, mode, w32)
            d = O.reprojection_rmse_delta(H, ref, p.src)
            diff = int((H != ref).sum())
            ulp = np.abs(H.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64)).max()
            print(f"{name} cells {vert.shape[0] * vert.shape[1]:6d} sums {mode} w32 {int(w32)}: rmse delta max {d.max():.3e} px, "
                  f"{diff} of {H.size} float32 differ, max {ulp} ulp, max |dH| {np.abs(H.astype(np.float64) - ref).max():.3e}")
