mkdir -p gpurun_out/r03m
export IRSDE_TUNING=1
for fl in 0 32768; do
for cfg in "IRSDE_WINO_FUSED64_NT=0" "IRSDE_WINO_FUSED64_NT=1"; do
  echo "== flags $fl $cfg" >> gpurun_out/r03m/ab.txt
  env $cfg timeout 200 python tools/op_profile.py 16 256 $fl > "gpurun_out/r03m/op_${fl}_$(echo $cfg | tr ' =' '__').txt" 2>&1
  grep -E "^total" "gpurun_out/r03m/op_${fl}_$(echo $cfg | tr ' =' '__').txt" >> gpurun_out/r03m/ab.txt
done; done
python tools/fused_nt_probe.py 10 > gpurun_out/r03m/nt_probe.txt 2>&1
cat gpurun_out/r03m/ab.txt gpurun_out/r03m/nt_probe.txt
