#!/usr/bin/env python3
"""What ONE rank of 8 pays for its rows of the C4 solve in one launch and in two (ShardedSolver's overlap of the first half's
all-gather with the second half's kernels): kernels only, one GPU playing rank 3."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from cvx_proj_amd import _native as N  # noqa: E402
from cvx_proj_amd.dist import hip_solve, row_partition  # noqa: E402
from cvx_proj_amd.synth import config_pair  # noqa: E402

p = config_pair(sys.argv[1] if len(sys.argv) > 1 else "C4", with_image=False)
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
rows, cols = p.vertices.shape[:2]
q = N.host_prepare(p.src, p.dst)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
table, den = t(N.host_build_table(p.src, q["cf1"], q["cf2"])), t(N.host_build_denorm(q["iC2"], q["C1"], q["iN2"], q["N1"]))
ra, rb = row_partition(rows, world)[min(3, world - 1)]
rm = ra + (rb - ra + 1) // 2
verts = p.vertices.reshape(-1, 2)
pieces = {"one launch": [(ra, rb)], "two launches": [(ra, rm), (rm, rb)]}
work = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
for label, pcs in pieces.items():
    vs = [t(verts[a * cols:b * cols]) for a, b in pcs]
    outs = [torch.empty((v.shape[0], 9), dtype=torch.float32, device=dev) for v in vs]
    run = lambda: [hip_solve(table, den, v, p.gamma, p.sigma, out=o, work=work) for v, o in zip(vs, outs)]  # noqa: E731
    for _ in range(300):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        run()
    torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter() - t0) / 300 * 1e6:.1f} us for {rb - ra} mesh rows of {cols} cells, {len(p.src)} keypoints")
