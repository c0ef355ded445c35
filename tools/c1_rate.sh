#!/bin/bash
# C1 (400 cells, fused launch) solve rate at several step counts: the per-step cost of a few-microsecond launch is
# sensitive to how many back-to-back launches the timed region holds.   tools/c1_rate.sh
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for S in 30 50 200 1000 5000; do
  for r in 1 2 3; do
    python $ROOT/bench.py --config C1 --no-cpu-baseline --no-cells --no-call-level --steps $S --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C1 steps=%-5d H/s=%.3e  %.2f us per solve' % ($S, d['value'], d['pairs']['solve_ms_per_step']*1e3))"
  done
done
