#!/bin/bash
# A/B of experimental BUILDS of the library on ONE box: every tools/variants/lib_*.so (built here with
# extra -D flags, e.g. -DAPAP_K1_GROUP=4; *.so travel to the GPU box, they are only git-ignored) against
# the in-tree library, ROUNDS alternating passes.  The working library is never overwritten: the
# binding loads $APAP_HIP_LIB.   tools/ab_build.sh [config]
CFG=${1:-C3}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for r in $(seq ${ROUNDS:-2}); do
  for L in "" $ROOT/tools/variants/lib_*.so; do
    APAP_HIP_LIB=$L python $ROOT/bench.py --config $CFG --steps 30 --no-cpu-baseline --no-cells --no-call-level 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('$CFG %-40s H/s=%.3e  assemble=%.1f us eigen=%.1f us' % ('${L##*/}' or 'in-tree', d['value'], k['assemble']*1e3, k['eigen']*1e3))"
  done
done
