"""K1 / K2 kernel times (HIP events of the context, steady state) for every solve form on one box, alternating passes:
variant (mfma = 16x16x4, mfma4 / mfma4x2 = 4x4x4_4b with 16 / 32 cells per wave) x APAP_OPT_MOMENTS (30 | 24) x
APAP_OPT_WEIGHTS_F32.   python tools/k1_modes.py [config] [rounds]"""
import ctypes
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from cvx_proj_amd import _native as N
from cvx_proj_amd.synth import config_pair

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
p = config_pair(cfg, with_image=False)
q = N.host_prepare(p.src, p.dst)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
tables = {m: t(N.host_build_table(p.src, q["cf1"], q["cf2"], moments=m)) for m in (30, 24)}
den = t(N.host_build_denorm(q["iC2"], q["C1"], q["iN2"], q["N1"]))
vert = t(p.vertices.reshape(-1, 2))
cells, n = vert.shape[0], len(p.src)
H = torch.zeros((cells, 9), dtype=torch.float32, device=dev)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
modes = [(v, m, w) for v in ("mfma", "mfma4", "mfma4x2") for (m, w) in ((30, 0), (24, 0), (24, 1))]
VAR = {"mfma": N.VARIANT_MFMA, "mfma4": N.VARIANT_MFMA4, "mfma4x2": N.VARIANT_MFMA4X2}
ref = None
for r in range(rounds):
    for v, m, w in modes:
        ctx = N.Context(variant=VAR[v], moments=m, weights_f32=w)
        need = max(N.lib().apap_solve_workspace_bytes(N._h(ctx), n, cells), 256)
        work = torch.empty(need, dtype=torch.uint8, device=dev)

        def solve():
            N.check(N.lib().apap_solve_device(N._h(ctx), tables[m].data_ptr(), n, vert.data_ptr(), cells, p.gamma, p.sigma, den.data_ptr(),
                                              H.data_ptr(), work.data_ptr(), need, stream))
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.25:       # sustained clocks
            for _ in range(20):
                solve()
            torch.cuda.synchronize()
        ctx.set("profile", 1)
        for _ in range(50):
            solve()
        torch.cuda.synchronize()
        prof = ctx.profile_read()
        k1, k2 = prof["assemble"][0] / 50 * 1e3, prof["eigen"][0] / 50 * 1e3
        Hh = H.cpu().numpy()
        if ref is None:
            ref = Hh.copy()
        diff = int((Hh != ref).sum())
        print(f"{cfg} {v:8s} moments {m} w32 {w}: assemble {k1:7.1f} us  eigen {k2:5.1f} us  H/s {cells / ((k1 + k2) * 1e-6):.3e}  "
              f"float32 values differing from the first form's grid {diff}/{Hh.size}", flush=True)
        ctx.close()
