#!/usr/bin/env python3
"""Static instruction mix of kernels in cvx_proj_amd/csrc/apap_kernels.gfx950.s (`make -C cvx_proj_amd/csrc asm`).
   tools/asm_count.py k_warp_fastILb0ELi4 k_warp_rowsILb0ELi4"""
import sys
from collections import Counter
import os
lines = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cvx_proj_amd/csrc/apap_kernels.gfx950.s")).read().split("\n")


def func(name):
    start = [i for i, l in enumerate(lines) if l.startswith("_ZN") and name in l.split(":")[0] and ": " in l][0]
    out = []
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith(".Lfunc_end"):
            break
        if not t or t.startswith((".", ";", "//")) or t.split(";")[0].strip().endswith(":"):
            continue
        out.append(t.split()[0])
    return out


for name in sys.argv[1:]:
    c = Counter(func(name))
    tot = sum(c.values())
    grp = lambda pre: sum(n for k, n in c.items() if k.startswith(pre))  # noqa: E731
    print(f"{name}: {tot} instructions, VALU {grp('v_')}, SALU {grp('s_')}, vector memory {grp(('global_', 'buffer_', 'flat_'))}, "
          f"f64 {sum(n for k, n in c.items() if 'f64' in k)}")
    print("   ", ", ".join(f"{k} {n}" for k, n in c.most_common(28)))
