#!/bin/bash
# tools/warp_pmc2.sh <tag> [config] : rocprofv3 PMC passes over tools/warp_forms.py (the gather kernel alone, warm) - memory-path
# counters beside the SQ ones: which unit is busiest during K3.  One pass per counter group (a block has few slots).
TAG=$1; CFG=${2:-C3}; FORMS=${3:-strips}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/wpmc2_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > "$OUT/avail.txt" 2>&1 || true
GROUPS_=(
 "GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max TA_BUSY_min"
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum"
 "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"
 "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
 "TCC_BUSY_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
 "TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_WRITEBACK_sum TCC_WRITE_sum TCC_READ_sum"
 "TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum TD_STORE_WAVEFRONT_sum TD_TC_STALL_sum"
)
i=0
for C in "${GROUPS_[@]}"; do
  i=$((i+1)); mkdir -p "$OUT/pmc_$i"
  timeout -k 10 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$i" -- python3 "$ROOT/tools/warp_forms.py" --config $CFG --steps 10 --cold-mb 0 --condition-s 0.01 --forms "$FORMS" > "$OUT/out_$i.txt" 2> "$OUT/err_$i.txt" || { echo "pass $i ($C) failed" >&2; tail -2 "$OUT/err_$i.txt" >&2; }
  echo "pass $i done" >> "$OUT/progress.txt"
done
python3 "$ROOT/tools/summarize_prof.py" "$OUT" | grep -E "k_warp_fast|k_warp_walk" > "$OUT/summary.txt"
cat "$OUT/summary.txt"
# keep the merged directory small
find "$OUT" -name "*.csv" -size +2M -delete
