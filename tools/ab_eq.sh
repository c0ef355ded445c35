#!/bin/bash
# A/B of experimental builds (tools/variants/lib_*.so) against the in-tree library for the callers of the path:
# equalisation and RANSAC times of bench.py's extras, HIP events.   tools/ab_eq.sh [config]
CFG=${1:-C3}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for r in $(seq ${ROUNDS:-3}); do
  for L in "" $ROOT/tools/variants/lib_*.so; do
    APAP_HIP_LIB=$L python $ROOT/bench.py --config $CFG --steps 30 --no-cpu-baseline --no-cells --no-call-level 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('$CFG %-24s eq_hist(+lut)=%.1f us eq_apply=%.1f us  equalize call=%.1f us  ransac=%.1f us' % ('${L##*/}' or 'in-tree', k['eq_hist']*1e3, k['eq_apply']*1e3, d['equalize']['ms_per_step']*1e3, k['ransac']*1e3))"
  done
done
