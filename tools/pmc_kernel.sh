#!/bin/bash
# Per-kernel SQ counters of ONE command on the GPU box (run through gpurun): one rocprofv3 --pmc pass per counter group,
# then tools/pmc_kernel.py prints the per-dispatch means for the kernels matching <regex>.
#   usage: bash tools/pmc_kernel.sh <tag> <kernel-regex> <command...>
set -u
TAG=$1; RE=$2; shift 2
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
GROUPS_=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
 "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
 "SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE"
 "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_SALU"
 "SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"
)
# (TA_* / TCP_* / TCC_* groups are not collected here: a TA_BUSY / TA_BUFFER_* pass aborted rocprofv3 on this pool and ran
#  into the time limit)
i=0
for G in "${GROUPS_[@]}"; do
  timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d "$OUT/g$i" -o pmc -- "$@" > /dev/null 2> "$OUT/g$i.err"
  i=$((i+1))
done
python "$REPO/tools/pmc_kernel.py" "$RE" $(find "$OUT" -name "*counter_collection.csv" | sort) > "$OUT/summary.txt"
rm -rf "$OUT"/g[0-9]
cat "$OUT/summary.txt"
