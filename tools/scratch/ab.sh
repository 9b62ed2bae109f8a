mkdir -p gpurun_out/r03p
export IRSDE_TUNING=1
for cfg in "IRSDE_SPLIT_NT=0" "IRSDE_SPLIT_NT=1"; do
  echo "== $cfg" >> gpurun_out/r03p/ab.txt
  env $cfg timeout 200 python tools/op_profile.py 16 256 32768 > "gpurun_out/r03p/op_$(echo $cfg | tr ' =' '__').txt" 2>&1
  grep -E "^total" "gpurun_out/r03p/op_$(echo $cfg | tr ' =' '__').txt" >> gpurun_out/r03p/ab.txt
done
cat gpurun_out/r03p/ab.txt
paste <(grep -E "gemm x36|wino_output" gpurun_out/r03p/op_IRSDE_SPLIT_NT_0.txt | awk '{print $1}') <(grep -E "gemm x36|wino_output" gpurun_out/r03p/op_IRSDE_SPLIT_NT_1.txt | cut -c1-90) | head -50
