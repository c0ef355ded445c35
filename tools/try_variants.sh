#!/bin/bash
# A/B harness: swap in experimental builds of the library (tools/variants/lib_*.so) on ONE box,
# print kernel times for ROUNDS alternating passes and, with PARITY=1, run the GPU parity
# suite against each variant.  The working library is restored at the end.
ROUNDS=${ROUNDS:-2}
cp cvx_proj_amd/libapap_hip.so /tmp/lib_base.so
for r in $(seq $ROUNDS); do
  for L in /tmp/lib_base.so tools/variants/lib_*.so; do
    cp $L cvx_proj_amd/libapap_hip.so
    python bench.py --steps 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('$L  H/s=%.3e  assemble=%.1f us eigen=%.1f setup=%.1f warp=%.1f us' % (d['value'], k['assemble']*1e3, k['eigen']*1e3, k['invert']*1e3, k['warp']*1e3))"
  done
done
if [ -n "$PARITY" ]; then
  for L in tools/variants/lib_*.so; do
    cp $L cvx_proj_amd/libapap_hip.so
    echo "== parity $L"
    python -m pytest tests -m gpu -q -x 2>&1 | tail -n 3
  done
fi
cp /tmp/lib_base.so cvx_proj_amd/libapap_hip.so
