#!/bin/bash
# steady-state C1 solve rate (1000 back-to-back fused launches), in-tree library against tools/variants/lib_*.so
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for r in $(seq ${ROUNDS:-3}); do
  for L in "" $ROOT/tools/variants/lib_*.so; do
    APAP_HIP_LIB=$L python $ROOT/bench.py --config C1 --no-cpu-baseline --no-cells --no-call-level --steps 1000 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C1 %-20s H/s=%.3e  %.3f us per solve' % ('${L##*/}' or 'in-tree', d['value'], d['pairs']['solve_ms_per_step']*1e3))"
  done
done
