#!/bin/bash
# tools/sweep.sh: kernel times of the K1 variants for several APAP_OPT_WANT_WAVES settings (C3 unless CONFIG set)
CFG=${CONFIG:-C3}
for V in valu mfma; do for W in 2048 4096 8192; do
  python bench.py --config $CFG --variant $V --want-waves $W --steps 20 --no-cpu-baseline --no-cells --no-call-level 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('$CFG $V want_waves=$W  H/s=%.3e  assemble=%.1f us eigen=%.1f us | invert=%.1f lut=%.1f warp=%.1f us  frac=%.3f' % (d['value'], k['assemble']*1e3, k['eigen']*1e3, k['invert']*1e3, k['lut']*1e3, k['warp']*1e3, d['roofline']['frac']))"
done; done
