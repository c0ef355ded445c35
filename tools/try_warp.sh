#!/bin/bash
# A/B of the warp kernel forms on one box: APAP_WARP_KERNEL = 0 (flat order) or rows per wave
for r in 1 2; do
for K in ${KERNELS:-0 2 4 8 16}; do
  APAP_WARP_KERNEL=$K python bench.py --steps 30 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('rows=$K  warp kernel %.1f us  warp step %.1f us (%.3e Mpix/s)  stitch step %.1f us' % (k['warp']*1e3, d['warp']['ms_per_step']*1e3, d['warp']['value'], d['stitch']['ms_per_step']*1e3))"
done
done
if [ -n "$PARITY" ]; then
  for K in ${PARITY_KERNELS:-2 4 8 16}; do
    echo "== parity rows=$K"; APAP_WARP_KERNEL=$K python -m pytest tests -m gpu -q -x 2>&1 | tail -n 3
  done
fi
