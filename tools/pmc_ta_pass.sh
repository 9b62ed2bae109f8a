#!/bin/bash
# rocprofv3 counter passes over the fused Winograd kernels' vector-memory path (TA / TD / TCP units), GPU box.   usage: bash tools/pmc_ta_pass.sh <tag> [variants] [sets: all|ta]
# (at most two counters of one unit per pass: three TA counters are "more than the hardware can collect")
set -u
TAG=${1:-ta}; VAR=${2:-430,432,448}; WHICH=${3:-all}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SETS=("TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum GRBM_GUI_ACTIVE" "TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE")
if [ "$WHICH" = all ]; then
  SETS+=("TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum GRBM_GUI_ACTIVE"
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE"
         "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE")
fi
i=0
for SET in "${SETS[@]}"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/q$i" -o pmc -- python $REPO/tools/wino_one_layer.py $VAR 1 > "$OUT/q$i.log" 2>&1
  grep -m1 "exceeds the capabilities" "$OUT/q$i.log"
done
python $REPO/tools/pmc_ta_summary.py "fused Winograd kernels, tools/wino_one_layer.py $VAR (B=16: 128->128 res @256^2, 512->512 res @64^2)" $(find "$OUT" -name "*counter_collection.csv") > "$OUT/pmc_ta_$WHICH.txt" 2>&1
cat "$OUT/pmc_ta_$WHICH.txt"
