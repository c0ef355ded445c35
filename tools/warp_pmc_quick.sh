#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/wpmc2_final
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" "TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max TA_BUSY_min" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1)); mkdir -p "$OUT/pmc_$i"
  timeout -k 10 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$i" -- python3 "$ROOT/tools/warp_forms.py" --config C3 --steps 10 --cold-mb 0 --condition-s 0.01 --forms "strips" > "$OUT/out_$i.txt" 2> "$OUT/err_$i.txt" || echo "pass $i failed"
done
python3 "$ROOT/tools/summarize_prof.py" "$OUT" | grep -E "k_warp_fast" > "$OUT/summary.txt"
cat "$OUT/summary.txt"
find "$OUT" -name "*.csv" -size +2M -delete
