"""r06: the split chain's safety net inside ONE process: two host threads launch the 4-groups-per-image kernel with 256 groups each on their own streams
(irsde_bench_naf_chain) — 512 groups on 256 CUs.  If the hardware runs the two kernels concurrently they can end up half resident each: every such launch must
give up (spin timeout -> error) and end; the script reports what happened and that the GPU still answers."""
import ctypes, sys, os, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from image_restoration_sde_amd import _lib
L = _lib.lib()
res = {}
def work(tag, iters):
    ok = bad = 0; t0 = time.time(); msgs = []
    for i in range(3):
        ms = ctypes.c_double()
        rc = L.irsde_bench_naf_chain(24, 28, 64, iters, ctypes.byref(ms))
        if rc == 0: ok += 1
        else: bad += 1; msgs.append(L.irsde_last_error().decode()[:90])
    res[tag] = (ok, bad, time.time() - t0, msgs)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ts = [threading.Thread(target=work, args=(k, iters)) for k in (1, 2)]
t0 = time.time()
for t in ts: t.start()
for t in ts: t.join()
for k, (ok, bad, dt, msgs) in sorted(res.items()):
    print("thread %d: %d calls ok, %d reported a co-residency timeout, %.1f s %s" % (k, ok, bad, dt, msgs[:1]))
print("wall %.1f s" % (time.time() - t0))
ms = ctypes.c_double()
print("afterwards alone: rc", L.irsde_bench_naf_chain(24, 28, 64, 5, ctypes.byref(ms)), "%.3f ms per launch" % ms.value)
