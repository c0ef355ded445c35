import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from cvx_proj_amd import _native as N
g = dict(np.load('tests/golden/tiny_sigma6.npz'))
fw, fh, ox, oy = (int(v) for v in g["final"])
for opts in (dict(warp_stage=2), dict(warp_stage=1), dict(warp_stage=2, warp_min_run=100000), dict(warp_stage=2, warp_min_run=1, warp_waves=32)):
    ctx = N.Context(warp_walk=1, **opts)
    w, hinv = N.local_warp(g["img"], g["H_ref"], g["mesh"][0], g["mesh"][1], fw, fh, ox, oy, ctx=ctx)
    d = (w != g["warped_ref"]).any(axis=-1)
    ys, xs = np.nonzero(d)
    print(opts, "diff pixels", d.sum(), "of", d.size, "rows", sorted(set(ys.tolist()))[:40], "cols", sorted(set(xs.tolist()))[:40])
    for y, x in list(zip(ys, xs))[:10]:
        print("  ", y, x, w[y, x], g["warped_ref"][y, x])
    ctx.close()
print("mesh_h", g["mesh"][1], "mesh_w", g["mesh"][0])
