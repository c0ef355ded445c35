#!/bin/bash
# tools/pmc.sh <tag> "<counters>" [bench.py args]: one rocprofv3 --pmc pass, prints per-kernel averages.
# rocprofv3 aborts AND then hangs in its signal handler when a pass asks for more counters of
# one block than the hardware has slots ("Request exceeds the capabilities of the hardware"),
# hence the hard timeout.
TAG=$1; C=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT/pmc_x"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_x" -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 3 --warmup 1 "$@" > /dev/null 2> "$OUT/err.txt" || { echo "pass [$C] failed:"; grep -m2 -E "exceeds|error code" "$OUT/err.txt"; exit 0; }
python3 "$ROOT/tools/summarize_prof.py" "$OUT" | grep -v "^=="
