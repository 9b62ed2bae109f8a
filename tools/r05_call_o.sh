#!/bin/bash
# r05 GPU call O: rocprofv3 round (kernel trace, HBM counters, MFMA counters) of BASELINE configs[2] (16 x 256^2, bf16_act, reverse_ode) on the final tree
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd "$REPO"
bash tools/profile_round.sh r05_final2_bf16_act --dtype bf16_act --mode ode > /dev/null 2>&1
ls gpurun_out/r05_final2_bf16_act
sed -n 1,40p gpurun_out/r05_final2_bf16_act/bench_pmc_hbm.txt | cut -c1-160
head -20 gpurun_out/r05_final2_bf16_act/bench_pmc_mfma.txt | cut -c1-200
