#!/bin/bash
# A/B of experimental BUILDS of the library (tools/variants/lib_*.so, see tools/ab_build.sh) on the WARP half:
# K3 time by HIP events, warp step, fused stitch step.   tools/ab_warp.sh [config]
CFG=${1:-C3}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for r in $(seq ${ROUNDS:-2}); do
  for L in "" $ROOT/tools/variants/lib_*.so; do
    APAP_HIP_LIB=$L python $ROOT/bench.py --config $CFG --steps 30 --no-cpu-baseline --no-cells --no-call-level ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('$CFG %-34s warp kernel %.1f us  set-up %.1f us  warp step %.1f us  stitch step %.1f us' % ('${L##*/}' or 'in-tree', k['warp']*1e3, k['invert']*1e3, d['warp']['ms_per_step']*1e3, d['stitch']['ms_per_step']*1e3))"
  done
done
