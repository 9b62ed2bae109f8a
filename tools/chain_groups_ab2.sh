# r06: the split chain kernel next to the concurrent-sub-batch sampler (configs[4] shapes), T = 100 (bench.py --model latent), one GPU box
mkdir -p gpurun_out/r06v
run() {  # B G SUB
  IRSDE_TUNING=1 IRSDE_NAF_CHAIN_SPLIT=$2 IRSDE_SUBBATCHES=$3 timeout 300 python bench.py --model latent --dtype fp16 --batch $1 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-live-pmc --no-profile > gpurun_out/r06v/b$1_g$2_s$3.json 2> gpurun_out/r06v/b$1_g$2_s$3.err
  echo "B=$1 groups=$2 sub-batches=$3 (0 = heuristic) $(grep -o '"value": *[0-9.]*' gpurun_out/r06v/b$1_g$2_s$3.json | head -1)"; grep -h "Error" gpurun_out/r06v/b$1_g$2_s$3.err | tail -1 | cut -c1-160
}
run 16 4 2
run 64 1 0
run 64 4 0
run 64 4 1
run 64 2 0
run 128 1 0
run 128 2 0
