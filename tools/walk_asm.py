#!/usr/bin/env python3
"""Memory operations, waits and register use of the k_warp_walk instantiations in csrc/apap_kernels.gfx950.s (make asm)."""
import re
import sys
s = open('cvx_proj_amd/csrc/apap_kernels.gfx950.s').read()
for m in re.finditer(r'\.amdhsa_kernel (_Z\w*k_warp_walk\w*)(.*?)\.end_amdhsa_kernel', s, re.S):
    d = m.group(2)
    print(m.group(1), 'vgpr', re.findall(r'next_free_vgpr (\d+)', d), 'sgpr', re.findall(r'next_free_sgpr (\d+)', d), 'scratch',
          re.findall(r'private_segment_fixed_size (\d+)', d))
which = sys.argv[1] if len(sys.argv) > 1 else 'ILb0ELi2E'
m = re.search(r'^_ZN12_GLOBAL__N_111k_warp_walk' + which + r'\w*:', s, re.M)
body = s[m.end():]
body = body[:body.index('.Lfunc_end')]
lines = body.splitlines()
print(sum(1 for l in lines if re.match(r'\s+v_', l)), 'valu', sum(1 for l in lines if re.match(r'\s+s_', l)), 'salu (static)')
for i, l in enumerate(lines):
    t = l.strip()
    if re.match(r'(buffer_load|global_load|global_store_dwordx3|s_waitcnt|s_load|s_endpgm|scratch_)', t):
        print(f"{i:5d} {t}")
