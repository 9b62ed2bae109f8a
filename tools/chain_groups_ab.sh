mkdir -p gpurun_out/r06aa
timeout 120 python -c "
import ctypes,sys
sys.path.insert(0,'.')
from image_restoration_sde_amd import _lib
L=_lib.probes_lib()
for B in (8,):
    ms=ctypes.c_double(); rc=L.irsde_bench_naf_chain(25,28,B,3,ctypes.byref(ms)); print('B=%d rc %d %.3f ms per launch'%(B,rc,ms.value),flush=True)
" 2>&1 | grep -v amdgpu | tee gpurun_out/r06aa/chain_stamps.txt
run() {
  IRSDE_TUNING=1 IRSDE_NAF_CHAIN_SPLIT=$2 timeout 300 python bench.py --model latent --dtype fp16 --batch $1 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-live-pmc --no-profile > gpurun_out/r06aa/b$1_g$2.json 2> gpurun_out/r06aa/b$1_g$2.err
  echo "B=$1 groups=$2: $(grep -o '"value": *[0-9.]*' gpurun_out/r06aa/b$1_g$2.json | head -1) img/s $(grep -h 'Error' gpurun_out/r06aa/b$1_g$2.err | tail -1 | cut -c1-100)"
}
for B in 8 16 32 64 128; do run $B 1; run $B 0; done 2>&1 | tee gpurun_out/r06aa/ab.txt
