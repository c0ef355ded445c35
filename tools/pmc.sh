#!/bin/bash
# tools/pmc.sh <tag> "<counters>" [bench.py args]: one rocprofv3 --pmc pass, prints per-kernel averages.
# rocprofv3 aborts AND then hangs in its signal handler when a pass asks for more counters of
# one block than the hardware has slots ("Request exceeds the capabilities of the hardware").
# So the request is checked BEFORE the launch: at most 8 counters of the SQ block and 1 each of
# FETCH_SIZE / WRITE_SIZE (they do not fit one pass together) - the sets tools/profile.sh uses -
# and a failed or timed-out pass exits non-zero.
TAG=$1; C=$2; shift 2
NSQ=$(echo $C | tr ' ' '\n' | grep -c '^SQ_'); NTC=$(echo $C | tr ' ' '\n' | grep -c -E '^(FETCH_SIZE|WRITE_SIZE)$')
if [ "$NSQ" -gt 8 ] || [ "$NTC" -gt 1 ]; then
  echo "pmc.sh: [$C] asks for $NSQ SQ counters (max 8 per pass) / $NTC of FETCH_SIZE+WRITE_SIZE (max 1): split it" >&2; exit 2
fi
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT/pmc_x"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_x" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-cells --no-call-level --steps 3 --warmup 1 "$@" > /dev/null 2> "$OUT/err.txt" || { echo "pass [$C] failed:" >&2; grep -m2 -E "exceeds|error code" "$OUT/err.txt" >&2; exit 1; }
python3 "$ROOT/tools/summarize_prof.py" "$OUT" | grep -v "^=="
