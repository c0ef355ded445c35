#!/bin/bash
# A/B of library builds x kernel forms on the gather kernel alone (tools/warp_forms.py): in-tree + every tools/variants/lib_*.so.
#   tools/ab_forms.sh [config] ["forms"]
CFG=${1:-C3}
FORMS=${2:-"strips;strips,warp_rows=8"}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for r in $(seq ${ROUNDS:-1}); do
  for L in "" $ROOT/tools/variants/lib_*.so; do
    APAP_HIP_LIB=$L python $ROOT/tools/warp_forms.py --config $CFG --steps 100 --forms "$FORMS" ${FORM_ARGS} 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception:
        print('   ', l.rstrip()[:200]); continue
    print('$CFG %-22s %-28s warm %6.2f us  cold %6.2f us' % ('${L##*/}' or 'in-tree', d['form'], d['warm_back_to_back_us'], d['cold_back_to_back_us']))"
  done
done
