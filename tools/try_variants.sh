#!/bin/bash
# swap in experimental builds of the library (tools/variants/lib_*.so) and print kernel times
cp cvx_proj_amd/libapap_hip.so /tmp/lib_base.so
for L in /tmp/lib_base.so tools/variants/lib_*.so; do
  cp $L cvx_proj_amd/libapap_hip.so
  python bench.py --steps 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('$L  H/s=%.3e  assemble=%.1f us eigen=%.1f setup=%.1f warp=%.1f us' % (d['value'], k['assemble']*1e3, k['eigen']*1e3, k['invert']*1e3, k['warp']*1e3))"
done
cp /tmp/lib_base.so cvx_proj_amd/libapap_hip.so
