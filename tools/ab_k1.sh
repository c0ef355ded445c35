#!/bin/bash
# A/B of the K1 variants on ONE box: kernel times from bench.py's HIP events, ROUNDS alternating passes.
#   tools/ab_k1.sh [config]       (ROUNDS=3 by default)
CFG=${1:-C3}
for r in $(seq ${ROUNDS:-3}); do
  for V in mfma mfma4 mfma4x2 valu; do
    python bench.py --config $CFG --steps 30 --no-cpu-baseline --no-cells --no-call-level --variant $V 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('$CFG %-8s H/s=%.3e  assemble=%.1f us eigen=%.1f us solve step %.1f us' % ('$V', d['value'], k['assemble']*1e3, k['eigen']*1e3, d['solve_ms_per_step']*1e3))"
  done
done
