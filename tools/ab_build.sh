#!/bin/bash
# A/B of experimental BUILDS of the library on ONE box: every tools/variants/lib_*.so (built from a modified
# copy of cvx_proj_amd/csrc - the product sources carry no tuning switches since round 6 -, e.g.
#   make -C /path/to/copy/csrc OUT=$PWD/tools/variants/lib_try.so
# ; tools/variants/ is git-ignored AND gpurun-ignored: build on the GPU box, in the same call) against
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
