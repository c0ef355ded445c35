#!/bin/bash
# tools/warp_pmc.sh <tag> "<forms>" : rocprofv3 PMC passes over tools/warp_forms.py (gather kernel alone, warm), SQ counters per launch.
# Two passes of <= 8 SQ counters each (rocprofv3 hangs when a pass asks for more than the block has slots).
TAG=$1; FORMS=$2; CFG=${3:-C3}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/wpmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM"
i=0
for C in "$P1" "$P2"; do
  i=$((i+1)); mkdir -p "$OUT/pmc_$i"
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$i" -- python3 "$ROOT/tools/warp_forms.py" --config $CFG --steps 10 --cold-mb 0 --condition-s 0.01 --forms "$FORMS" > "$OUT/out_$i.txt" 2> "$OUT/err_$i.txt" || { echo "pass $i failed" >&2; tail -3 "$OUT/err_$i.txt" >&2; }
done
python3 "$ROOT/tools/summarize_prof.py" "$OUT" | grep -E "k_warp_fast|k_warp_walk"
