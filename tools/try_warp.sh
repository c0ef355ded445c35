#!/bin/bash
# A/B of the warp kernel forms on one box: fast (float32 estimate) on / off x rows per wave (APAP_OPT_WARP_ROWS, 0 = flat order)
for r in 1 2; do
for F in ${FAST:-1 0}; do
for K in ${KERNELS:-2 4 8}; do
  python bench.py --steps 30 --warp-rows $K --warp-fast $F --no-cpu-baseline --no-cells --no-call-level ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('fast=$F rows=$K  warp kernel %.1f us  warp step %.1f us (%.3e Mpix/s)  stitch step %.1f us' % (k['warp']*1e3, d['warp']['ms_per_step']*1e3, d['warp']['value'], d['stitch']['ms_per_step']*1e3))"
done
done
done
