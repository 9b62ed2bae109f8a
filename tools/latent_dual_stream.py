"""configs[4] latent pipeline: one batch of 64 on one engine vs two half batches on two engines / two streams (GPU box).
The per-image NAFBlock chain kernel occupies one CU per image (64 of 256 at B = 64); the question is whether a second engine's
launches fill the idle CUs.  usage: python tools/latent_dual_stream.py [T]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import image_restoration_sde_amd as P

T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
base = [w for w in bench.SECONDARY if w["model"] == "latent"][0]

def mk(batch, rank):
    w = dict(base); w["batch"] = batch; w["T"] = T
    return bench.Workload(P, w, dev, rank, 1, "weak")

def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

w64 = mk(64, 0)
t64 = timed(w64.one_step)
print("one engine, B = 64: %.1f ms per call -> %.1f img/s" % (t64 * 1e3, 64 / t64))
for parts in (2, 4):
    ws = [mk(64 // parts, r) for r in range(parts)]
    ss = [torch.cuda.Stream() for _ in range(parts)]
    def run():
        for w, s in zip(ws, ss):
            with torch.cuda.stream(s):
                w.one_step()
    tp = timed(run)
    print("%d engines x B = %d on %d streams, one host thread: %.1f ms per call -> %.1f img/s" % (parts, 64 // parts, parts, tp * 1e3, 64 / tp))
    def run_threads():
        def body(w, s):
            with torch.cuda.stream(s):
                w.one_step()
        th = [threading.Thread(target=body, args=(w, s)) for w, s in zip(ws, ss)]
        for t in th: t.start()
        for t in th: t.join()
    tp = timed(run_threads)
    print("%d engines x B = %d on %d streams, one host thread each: %.1f ms per call -> %.1f img/s" % (parts, 64 // parts, parts, tp * 1e3, 64 / tp))
    ts = timed(ws[0].one_step)
    print("   one of them alone: %.1f ms -> %.1f img/s" % (ts * 1e3, (64 // parts) / ts))
    del ws
