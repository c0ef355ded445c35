#!/bin/bash
# Collect the rocprofv3 evidence for one bench.py configuration on the GPU box.
#   tools/profile.sh <tag> [bench.py args...]
# Pass --no-c5 for clean per-kernel averages of the headline configuration: the `pairs` object launches the same kernels over 64
# pairs at once (k_warp_fast, k_assemble_mfma, k_eigen_denorm, k_warp_setup with grid.z = pair), and rocprofv3 averages by name.
# Writes gpurun_out/prof_<tag>/{stats,pmc_*}; copy the summaries you want judged to profiles/.
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace + stats (per-kernel average duration)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-call-level --no-cells --steps 20 "$@" > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err" || { tail -5 "$OUT/stats.err"; exit 1; }
# 2. PMC passes, each in its own run (FETCH_SIZE and WRITE_SIZE do not fit one pass)
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$N" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-cells --no-call-level --steps 3 --warmup 1 --condition-ms 0 "$@" > /dev/null 2> "$OUT/pmc_$N.err" || { echo "pmc pass $C failed"; tail -3 "$OUT/pmc_$N.err"; }
done
python3 "$ROOT/tools/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
# the traces hold one line per launch (tens of thousands with the bench's clock conditioning): keep the summaries only
find "$OUT" -name "*_kernel_trace.csv" -delete -o -name "*_counter_collection.csv" -delete
cat "$OUT/summary.txt"
