#!/bin/bash
# A/B of library builds on the batched warp (gather phase only): tools/ab_batch.sh [config] [pairs]
CFG=${1:-C5}; NB=${2:-32}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for r in $(seq ${ROUNDS:-1}); do
  for L in "" $ROOT/tools/variants/lib_*.so; do
    APAP_HIP_LIB=$L python $ROOT/tools/batch_warp_rate.py --configs $CFG --batches $NB --steps 20 --phases 4 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$CFG x%d %-22s %6.2f us per pair' % (d['pairs_per_launch'], '${L##*/}' or 'in-tree', d['us_per_pair']))"
  done
done
