"""First-light diagnostics on a GPU box: prints (never asserts) per-kernel / per-layer errors of the HIP path
against the oracle, plus a first timing of the full-size network evaluation.  Output goes to stdout."""
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import image_restoration_sde_amd as P  # noqa: E402
from image_restoration_sde_amd import _lib  # noqa: E402
from oracle import irsde_oracle as O  # noqa: E402

DEV = "cuda:0"


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


def section(name):
    print("\n==== %s ====" % name, flush=True)


def main():
    print(torch.__version__, torch.cuda.get_device_name(0), "cpus", os.cpu_count(), flush=True)
    section("conv kernel cases (MFMA / naive / split-K vs float64 oracle)")
    try:
        import test_gpu_parity as T
        for name, case in T.CONV_CASES.items():
            B, C0, C1, H, W, Cout, K, stride, pad, in_shift, has_bias, has_film, silu, has_res = case
            rs = np.random.RandomState(1)
            x0 = rs.standard_normal((B, C0, H, W)).astype(np.float32)
            x1 = rs.standard_normal((B, C1, H, W)).astype(np.float32) if C1 else None
            w = (rs.standard_normal((Cout, C0 + C1, K, K)) / np.sqrt((C0 + C1) * K * K)).astype(np.float32)
            bias = rs.standard_normal(Cout).astype(np.float32) if has_bias else None
            film = (0.3 * rs.standard_normal((1, 2 * Cout))).astype(np.float32) if has_film else None
            Ho = ((H << in_shift) + 2 * pad - K) // stride + 1
            Wo = ((W << in_shift) + 2 * pad - K) // stride + 1
            res = rs.standard_normal((B, Cout, Ho, Wo)).astype(np.float32) if has_res else None
            ref = T.oracle_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res)
            errs = []
            for kw in (dict(), dict(naive=1), dict(splits=3)):
                try:
                    got = T.run_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res, **kw)
                    errs.append("%.2e" % rel(got, ref))
                except Exception as ex:
                    errs.append("EXC %r" % (ex,))
            print("%-24s mfma/naive/splitk: %s" % (name, errs), flush=True)
    except Exception:
        traceback.print_exc()

    section("per-layer taps, nf=32 depth=2, 2x3x24x20, t=7 (vs float64 oracle)")
    try:
        nf, depth, B, H, W = 32, 2, 2, 24, 20
        params = O.synth_params(seed=0, nf=nf, depth=depth)
        m = P.ConditionalUNet(3, 3, nf, depth=depth)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
        m.engine_flags = _lib.FLAG_KEEP_ACTIVATIONS
        m = m.to(DEV).eval()
        lq, xT = O.synth_inputs(1234, B, H, W)
        taps = {}
        ref = O.unet_forward(params, xT, lq, 7, depth=depth, dtype=np.float64, taps=taps)
        y = m(torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV), 7).cpu().numpy()
        for name, want in taps.items():
            if name == "time_emb":
                continue
            try:
                got = m.debug_tap(name).numpy()
                print("%-18s shape %-18s rel err %.2e" % (name, got.shape, rel(got, want)), flush=True)
            except Exception as ex:
                print(name, "EXC", ex)
        print("output rel err %.2e" % rel(y, ref))
    except Exception:
        traceback.print_exc()

    section("nf=64 depth=4 forward vs reference golden")
    try:
        g = np.load(os.path.join(ROOT, "tests/golden/forward.npz"))
        params = O.synth_params(seed=0, nf=64, depth=4)
        m64 = P.ConditionalUNet(3, 3, 64, depth=4)
        m64.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
        m64 = m64.to(DEV).eval()
        for tag in ("nf64d4_1x64x64", "nf64d4_2x40x56"):
            nf, depth, B, H, W = (int(v) for v in g[tag + "/cfg"])
            lq, xT = O.synth_inputs(1234, B, H, W)
            for t in g[tag + "/ts"]:
                y = m64(torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV), int(t)).cpu().numpy()
                print(tag, int(t), "rel err %.2e" % rel(y, g[tag + "/t%d" % t]), flush=True)
    except Exception:
        traceback.print_exc()

    section("sampler nf64d4 1x32x32 T=100 vs reference golden (eager / graph)")
    try:
        g = np.load(os.path.join(ROOT, "tests/golden/sampler.npz"))
        tag = "nf64d4_1x32x32_T100"
        nf, depth, B, H, W, T = (int(v) for v in g[tag + "/cfg"])
        lq, xT = O.synth_inputs(1234, B, H, W)
        z = O.synth_noise(7, T, (B, 3, H, W))
        sde = P.IRSDE(10, T, "cosine", 0.005, device=DEV)
        sde.set_model(m64)
        sde.set_mu(torch.from_numpy(lq).to(DEV))
        sde.injected_noise = torch.from_numpy(z).to(DEV)
        for graph in (False, True):
            sde.use_graph = graph
            for mode, fn in (("sde", sde.reverse_sde), ("ode", sde.reverse_ode), ("posterior", sde.reverse_posterior)):
                torch.cuda.synchronize()
                t0 = time.time()
                y = fn(torch.from_numpy(xT).to(DEV)).cpu().numpy()
                print("graph=%d %-9s rel err %.2e  (%.2f s)" % (graph, mode, rel(y, g[tag + "/" + mode]), time.time() - t0),
                      flush=True)
    except Exception:
        traceback.print_exc()

    section("timing: one network evaluation, nf=64 depth=4")
    try:
        for B, S in ((1, 128), (1, 256), (4, 256), (16, 256)):
            lq, xT = O.synth_inputs(1, B, S, S)
            x, c = torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV)
            m64(x, c, 50)
            torch.cuda.synchronize()
            n = 3
            t0 = time.time()
            for _ in range(n):
                m64(x, c, 50)
            torch.cuda.synchronize()
            dt = (time.time() - t0) / n
            fl = 7.62e11 * B * (S * S) / 65536.0
            print("B=%d %dx%d: %.2f ms per evaluation  -> %.1f TFLOP/s (fp32 peak 157.3)" % (B, S, S, dt * 1e3, fl / dt / 1e12),
                  flush=True)
    except Exception:
        traceback.print_exc()

    section("profiled sampler steps, B=16 256x256, 3 steps")
    try:
        B, S, T = 16, 256, 100
        lq, xT = O.synth_inputs(1, B, S, S)
        sde = P.IRSDE(10, T, "cosine", 0.005, device=DEV)
        sde.set_model(m64)
        sde.set_mu(torch.from_numpy(lq).to(DEV))
        sde.profile = True
        sde.reverse_sde(torch.from_numpy(xT).to(DEV), T=3)
        torch.cuda.synchronize()
        pr = sde.last_profile()
        print(pr)
        print("conv TFLOP/s %.1f ; conv share of wall %.3f" % (pr["conv_flops"] / pr["conv_ms"] / 1e9, pr["conv_ms"] / pr["wall_ms"]))
    except Exception:
        traceback.print_exc()


if __name__ == "__main__":
    main()
