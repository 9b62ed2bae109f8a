#!/bin/bash
# r06: the safety net of the split NAFBlock chain.  Two processes each launch the 4-groups-per-image kernel with 256 groups (64 images) on ONE GPU at the same time:
# 512 groups cannot be co-resident on 256 CUs, so launches may find themselves half resident — they must give up after their spin limit (error word -> "spin timeout"
# from irsde_bench_naf_chain) and END, never hang.  Every process runs under `timeout`; prints what each one saw and the wall time.
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; mkdir -p gpurun_out/r06stress
run() { timeout 150 python - "$1" <<'PY' > gpurun_out/r06stress/p$1.txt 2>&1
import ctypes, sys, time
sys.path.insert(0, ".")
from image_restoration_sde_amd import _lib
L = _lib.lib()
t0 = time.time(); ok = bad = 0
for i in range(6):
    ms = ctypes.c_double()
    rc = L.irsde_bench_naf_chain(24, 28, 64, 2000, ctypes.byref(ms))
    if rc == 0: ok += 1
    else:
        bad += 1; print("call %d: rc %d %s" % (i, rc, L.irsde_last_error().decode()[:100]))
print("process %s: %d calls ok, %d reported a co-residency timeout, %.1f s" % (sys.argv[1], ok, bad, time.time() - t0))
PY
}
python -c "import time; open('gpurun_out/r06stress/t0','w').write(str(time.time()))"
run 1 & run 2 & run 3 &
wait
cat gpurun_out/r06stress/p1.txt gpurun_out/r06stress/p2.txt gpurun_out/r06stress/p3.txt | grep -v amdgpu
python -c "import time; print('wall %.1f s' % (time.time() - float(open('gpurun_out/r06stress/t0').read())))"
# the GPU must still answer
timeout 60 python -c "
import ctypes,sys
sys.path.insert(0,'.')
from image_restoration_sde_amd import _lib
ms=ctypes.c_double(); print('afterwards alone: rc', _lib.lib().irsde_bench_naf_chain(24,28,64,5,ctypes.byref(ms)), '%.3f ms per launch'%ms.value)" 2>&1 | grep -v amdgpu
