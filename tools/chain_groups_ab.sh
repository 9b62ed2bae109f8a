# r06: configs[4] shapes (bench.py --model latent --dtype fp16, T = 100) by work-groups per image of the NAFBlock chain and by concurrent sub-batches, one GPU box
mkdir -p gpurun_out/r06y
run() {  # B G SUB
  IRSDE_TUNING=1 IRSDE_NAF_CHAIN_SPLIT=$2 IRSDE_SUBBATCHES=$3 timeout 300 python bench.py --model latent --dtype fp16 --batch $1 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-live-pmc --no-profile > gpurun_out/r06y/b$1_g$2_s$3.json 2> gpurun_out/r06y/b$1_g$2_s$3.err
  echo "B=$1 groups=$2 sub-batches=$3 (0 = heuristic): $(grep -o '"value": *[0-9.]*' gpurun_out/r06y/b$1_g$2_s$3.json | head -1) img/s  $(grep -h 'Error' gpurun_out/r06y/b$1_g$2_s$3.err | tail -1 | cut -c1-120)"
}
for B in 8 16 32; do for G in 1 2 4; do run $B $G 0; done; done
run 64 1 0; run 64 2 0; run 64 4 0; run 64 1 1; run 64 2 1; run 64 4 1
run 128 1 0; run 128 2 0
