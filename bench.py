"""bench.py — images/s of the full T-step IR-SDE reverse sampler (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A "step" = one pass of the hot path over one batch: `IRSDE.reverse_sde` (T=100 network evaluations + state
updates) on a synthetic batch of 16 x 3 x 256 x 256 per GPU (BASELINE.json configs[1]); weak scaling: every
rank samples its own 16 images (global image index keyed RNG), the only collective is the final all_gather
of the restored images (RCCL over xGMI), which is inside the timed region.

`python bench.py --gpus N` with N > 1 and no torchrun environment launches itself: it re-executes under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one rank per GPU).

`value` is timed on the production path: one captured hipGraph per step replayed T times (`sde.use_graph`), no
per-kernel events.  After the timed region rank 0 runs ONE more, untimed, event-instrumented pass (eager launches
with a hipEvent pair around every kernel on the engine's stream) to fill `roofline`:
    roofline.frac = FLOPs the MFMA pipe executed (Winograd layers counted at their reduced multiply count)
                    / (time of the MFMA kernels + the Winograd transform kernels that belong to them) / peak   (<= 1)
`mfma_kernel_frac` is the same over the MFMA kernels alone; the direct-convolution-equivalent rate (which exceeds the
fp32 peak because Winograd skips multiplies) is reported as `algorithmic_equiv_TFLOPs`, never as a fraction.
`cpu_baseline`: the torch-CPU port of the reference timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3   # MI355X fp32 matrix = vector peak (guides/MI355X_MICROARCH.md)
PEAK_HBM_GBPS = 8000.0
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak (same guide)


def host_cores():
    """CPU cores this process may actually use: min(affinity, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def synth_state_dict(module, seed):
    """Deterministic synthetic weights straight from the module's own parameter shapes (values do not affect speed):
    conv / linear ~ U(+-1/sqrt(fan_in)) like PyTorch's default init, LayerNorm gains ~ U(0.5, 1.5), NAFNet beta / gamma
    ~ U(-0.5, 0.5) (zero in the reference, which would turn every block into the identity)."""
    import numpy as np
    import torch
    rs = np.random.RandomState(seed)
    sd = module.state_dict()
    out = {}
    for name in sorted(sd):
        shp = tuple(sd[name].shape)
        if name.endswith(".g"):
            a = rs.uniform(0.5, 1.5, size=shp)
        elif name.endswith("beta") or name.endswith("gamma"):
            a = rs.uniform(-0.5, 0.5, size=shp)
        else:
            wshape = tuple(sd[name[:-4] + "weight"].shape) if name.endswith("bias") else shp
            bound = 1.0 / math.sqrt(int(np.prod(wshape[1:])))
            a = rs.uniform(-bound, bound, size=shp)
        out[name] = torch.from_numpy(a.astype(np.float32))
    return out


def synth_inputs(seed, B, H, W, max_sigma):
    """LQ ~ U[0,1), x_T = LQ + N(0,1) * max_sigma / 255  (IRSDE.noise_state, sde_utils.py:360-361)."""
    import numpy as np
    rs = np.random.RandomState(seed)
    lq = rs.rand(B, 3, H, W).astype(np.float32)
    xT = (lq + rs.standard_normal(lq.shape).astype(np.float32) * np.float32(max_sigma / 255.0)).astype(np.float32)
    return lq, xT


def cpu_baseline(size, T, budget_s=20.0):
    """Reference CPU path (torch CPU ops, oracle/torch_cpu_port.py) on a bounded sample: B=1 image of
    size x size, as many reverse_sde steps as fit in ~budget_s (min 2, max 10), extrapolated to T steps.
    The only place bench.py touches oracle/ (the reported baseline; never the thing measured as `value`)."""
    import torch
    from oracle import irsde_oracle as O
    from oracle import torch_cpu_port as TP
    cores = host_cores()
    torch.set_num_threads(cores)
    params = {k: torch.from_numpy(v) for k, v in O.synth_params(seed=0, nf=64, depth=4).items()}
    lq, xT = O.synth_inputs(1234, 1, size, size)
    sch = O.irsde_schedule(10, T, "cosine", 0.005)
    z = torch.from_numpy(O.synth_noise(7, T, (1, 3, size, size)))
    x, mu = torch.from_numpy(xT), torch.from_numpy(lq)
    TP.reverse_sde_steps(params, sch, x, mu, z, T, 1)  # warm-up (oneDNN primitive creation)
    n, t0 = 0, time.time()
    while n < 10 and (n < 2 or time.time() - t0 < budget_s):
        x = TP.reverse_sde_steps(params, sch, x, mu, z, T - n, 1)
        n += 1
    per_step = (time.time() - t0) / n
    return {"value": 1.0 / (per_step * T), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "B=1 %dx%d, %d of T=%d reverse_sde steps timed (%.2f s/step), extrapolated x%d; torch %s CPU, "
                      "%d threads (affinity/cgroup quota; host has %d logical CPUs). kind=port: oracle/torch_cpu_port.py, the "
                      "reference's algorithm restated in torch-CPU functional ops and golden-checked against the reference "
                      "(tests/test_oracle_golden.py) - /root/reference itself does not exist on the GPU box.  The unmodified "
                      "reference classes measured in the build container (SURVEY.md 8d): 0.034 images/s at 1x128x128, T=100, 8 vCPU"
                      % (size, size, n, T, per_step, T, torch.__version__, cores, os.cpu_count() or 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU (BASELINE config 2: 16)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--T", type=int, default=100)
    ap.add_argument("--mode", default="sde", choices=["sde", "ode", "posterior"])
    ap.add_argument("--model", default="unet", choices=["unet", "nafnet", "dsde", "latent"],
                    help="unet: IR-SDE ConditionalUNet (BASELINE configs[1]); nafnet: Refusion ConditionalNAFNet (configs[3])")
    ap.add_argument("--max-sigma", type=float, default=None)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16", "bf16_act", "fp16"],
                    help="bf16 = BASELINE configs[2] (conv operands bf16, fp32 accumulate); fp16 = configs[4] (IEEE fp16 operands); "
                         "the headline metric is fp32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the untimed event-instrumented pass (no `roofline` object)")
    a = ap.parse_args()

    if a.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torchrun: launch one rank per GPU ourselves (the driver may call `python bench.py --gpus 8` directly)
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    import numpy as np
    import torch
    import torch.distributed as dist
    import image_restoration_sde_amd as P

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        sys.exit("bench.py: --gpus %d does not match WORLD_SIZE=%d of the launcher" % (a.gpus, world))
    # test hook (tests/test_gpu_parity.py on a 1-GPU box): IRSDE_BENCH_OVERSUBSCRIBE=1 lets several ranks share the visible
    # GPUs; RCCL refuses two ranks on one device, so the final gather then goes through gloo (host staging)
    oversub = os.environ.get("IRSDE_BENCH_OVERSUBSCRIBE") == "1" and torch.cuda.device_count() >= 1
    if torch.cuda.device_count() <= local_rank and not oversub:
        sys.exit("bench.py: rank %d needs GPU %d but only %d device(s) are visible" % (rank, local_rank, torch.cuda.device_count()))
    dev_index = local_rank % torch.cuda.device_count() if oversub else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if oversub:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    latent_model = None
    if a.model == "latent":  # latent-bokeh/options/bokeh/test/refusion.yml: UNet ch 64 [1,2,4] embed 4 (256^2 -> 64x64x4) + NAFNet
        ncfg = dict(width=64, enc_blk_nums=[1, 1, 1, 28], middle_blk_num=1, dec_blk_nums=[1, 1, 1, 1])
        model = P.latent_bokeh.ConditionalNAFNet(img_channel=4, **ncfg)   # lens-conditioned (lens_info kwargs)
        latent_model = P.latent.UNet(in_ch=3, out_ch=3, ch=64, ch_mult=[1, 2, 4], embed_dim=4)
        latent_model.load_state_dict(synth_state_dict(latent_model, 1))
        latent_model = latent_model.to(dev).eval()
    elif a.model == "dsde":  # denoising-sde/options/test/ir-sde.yml: unconditional UNet + DenoisingSDE(max_sigma=75, T=100)
        model = P.denoising_sde.ConditionalUNet(3, 3, 64, depth=4)
    elif a.model == "nafnet":  # refusion.yml network_G
        ncfg = dict(width=64, enc_blk_nums=[1, 1, 1, 28], middle_blk_num=1, dec_blk_nums=[1, 1, 1, 1])
        model = P.ConditionalNAFNet(img_channel=3, **ncfg)
    else:
        nf, depth = 64, 4
        model = P.ConditionalUNet(3, 3, nf, depth=depth)
    model.load_state_dict(synth_state_dict(model, 0))
    model = model.to(dev).eval()
    model.set_compute_dtype(a.dtype)
    max_sigma = a.max_sigma if a.max_sigma is not None else {"nafnet": 50, "dsde": 75, "latent": 50}.get(a.model, 10)
    if a.model == "dsde":
        sde = P.DenoisingSDE(max_sigma=max_sigma, T=a.T, device=dev)
    else:
        sde = P.IRSDE(max_sigma=max_sigma, T=a.T, schedule="cosine", eps=0.005, device=dev)
    sde.set_model(model)
    sde.seed = 7
    sde.profile = False    # the timed region runs the production path: hipGraph replay, no per-kernel events
    sde.use_graph = True
    want_profile = (not a.no_profile) and a.model not in ("dsde", "latent")

    nglobal = a.batch * world
    # every rank materialises only its shard of the synthetic global batch (same generator => same images)
    lq, xT = synth_inputs(1234, a.batch, a.size, a.size, max_sigma)
    rs = np.random.RandomState(1000 + rank)
    lq = np.clip(lq + 0.01 * rs.standard_normal(lq.shape).astype(np.float32), 0, 1)
    mu = torch.from_numpy(lq).to(dev)
    x_T = torch.from_numpy(xT).to(dev)
    sde.image_offset = rank * a.batch
    n_evals = a.T
    if a.model == "dsde":  # denoising-sde/test.py:103-107: reverse_ode from the optimal timestep of the noise level (sigma 25)
        n_evals = int(sde.get_optimal_timestep(25))
        fn = lambda x: sde.reverse_ode(x, T=n_evals) if a.mode != "sde" else sde.reverse_sde(x, T=n_evals)  # noqa: E731
    elif a.model == "latent":  # latent-dehazing/test.py:90-100: encode once, sample in the latent, decode once
        sample = {"sde": sde.reverse_sde, "ode": sde.reverse_ode, "posterior": sde.reverse_posterior}[a.mode]

        lens_rs = np.random.RandomState(5 + rank)   # per image: src_lens, tgt_lens, disparity (latent-bokeh/test.py:91)
        lens_info = [torch.from_numpy(lens_rs.uniform(lo, hi, a.batch).astype(np.float32)) for lo, hi in ((1.4, 2.8), (1.8, 16.0), (0.0, 100.0))]

        def fn(_unused):
            latent_LQ, hidden = latent_model.encode(mu)
            sde.set_mu(latent_LQ)
            return latent_model.decode(sample(sde.noise_state(latent_LQ), lens_info=lens_info), hidden)
    else:
        sde.set_mu(mu)
        fn = None

    def one_step():
        if fn is not None:
            out = fn(x_T)
            return P.gather_batch(out, nglobal) if world > 1 else out
        # the product's N>1 path: sample this rank's shard, then the final all_gather (the only collective)
        sde.image_offset = 0
        return P.sample_shard(sde, a.mode, x_T, mu, rank * a.batch, nglobal)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        out = one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = one_step()
    fence()
    dt = time.perf_counter() - t0
    assert out.shape[0] == nglobal and bool(torch.isfinite(out).all())
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    prof = None
    if want_profile and rank == 0:
        # untimed: the same sampling call once more with a hipEvent pair around every kernel (eager launches)
        sde.profile = True
        sde.image_offset = 0
        sde.set_mu(mu)
        {"sde": sde.reverse_sde, "ode": sde.reverse_ode, "posterior": sde.reverse_posterior}[a.mode](x_T)
        torch.cuda.synchronize()
        prof = sde.last_profile()
        sde.profile = False

    if rank == 0:
        imgs = nglobal * a.steps
        res = {
            "metric": ("restored images/sec at %dx%d, %d-step DenoisingSDE reverse sampler (optimal timestep of sigma=25)" % (a.size, a.size, n_evals)
                       if a.model == "dsde" else "restored images/sec at %dx%d, %d-step IR-SDE reverse sampler" % (a.size, a.size, a.T)),
            "value": imgs / dt, "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1000.0 * dt / a.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"fp32": "f32", "bf16": "bf16 operands / f32 accumulate+state",
                                          "bf16_act": "bf16 operands + bf16 activation storage / f32 accumulate+state",
                                          "fp16": "f16 operands / f32 accumulate+state"}[a.dtype], "data": "synthetic",
            "config": {"workload": ("denoising-sde unconditional UNet nf=64 depth=4 (full attention at the bottleneck), DenoisingSDE reverse_%s from the "
                                    "optimal timestep of sigma=25 (" + str(n_evals) + " network evaluations), batch=%d/GPU %dx%d, schedule T=%d, fp32"
                                    if a.model == "dsde" else "Latent-Refusion (latent-bokeh): latent UNet ch=64 [1,2,4] embed 4 (encode + decode once per image) + lens-conditioned ConditionalNAFNet "
                                    "width=64 enc[1,1,1,28] on the 64x64x4 latent, reverse_%s, batch=%d/GPU %dx%d, T=%d (BASELINE.json configs[4] shape)"
                                    if a.model == "latent" else "Refusion ConditionalNAFNet width=64 enc[1,1,1,28], reverse_%s, batch=%d/GPU %dx%d, T=%d, fp32 "
                                    "(BASELINE.json configs[3] network)" if a.model == "nafnet" else
                                    "IR-SDE deraining ConditionalUNet nf=64 depth=4, reverse_%s, batch=%d/GPU %dx%d, "
                                    "T=%d, fp32 (BASELINE.json configs[1])") % (a.mode, a.batch, a.size, a.size, a.T),
                       "compute_dtype": a.dtype, "global_batch": nglobal, "parallelism": "batch-shard x%d (no collective in the T loop; final all_gather over %s)"
                                      % (world, "gloo, ranks sharing GPUs: TEST HOOK, not a scaling number" if (world > 1 and oversub) else "RCCL")},
        }
        if prof is not None and prof["conv_ms"] > 0:
            # `achieved` = FLOPs the MFMA pipe EXECUTED (Winograd F(4x4,3x3) / F(2x2,3x3) layers issue 4x / 2.25x fewer
            # multiplies than the direct-convolution count of SURVEY.md 8d) / time of the MFMA kernels plus the Winograd
            # transform kernels that exist only to serve them: a true fraction of the fp32 MFMA roof (<= 1).
            conv_t = (prof["conv_ms"] + prof["wino_ms"]) * 1e-3
            ach = prof["conv_exec_flops"] / conv_t / 1e12
            exe = prof["conv_exec_flops"] / (prof["conv_ms"] * 1e-3) / 1e12
            alg = prof["conv_flops"] / conv_t / 1e12
            # HBM bytes per conv launch: rocprofv3 PMC passes cannot run inside this process; the figure comes from the
            # builder's PMC run of this same command, committed under profiles/ (source named next to it)
            traffic, traffic_src = None, None
            try:
                if a.batch == 16 and a.size == 256 and a.dtype == "fp32" and a.model == "unet":
                    pm = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_bench_pmc_hbm.json"))
                    traffic = json.load(open(os.path.join(ROOT, "profiles", pm[-1])))["traffic_bytes_per_launch"]
                    traffic_src = "profiles/%s (builder's rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; not measured in this run)" % pm[-1]
            except (OSError, IndexError, KeyError, ValueError):
                traffic, traffic_src = None, None
            peak = PEAK_FP32_TFLOPS if a.dtype == "fp32" else PEAK_BF16_TFLOPS
            res["roofline"] = {
                "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                "frac": ach / peak, "mfma_kernel_frac": exe / peak, "traffic": traffic, "traffic_source": traffic_src,
                "traffic_unit": "bytes per conv launch (2 x FETCH_SIZE + WRITE_SIZE, rocprofv3 PMC passes, profiles/*_bench_pmc_hbm.txt)",
                "algorithmic_equiv_TFLOPs": alg,
                "timing": "separate untimed pass after the timed region: eager launches, hipEvent pair around every kernel on the engine stream",
                "algorithmic_bytes_per_launch": prof["conv_bytes"] / max(prof["conv_launches"], 1),
                "kernel": "conv_igemm_kernel + gemm_zloop_kernel (fp32 v_mfma_f32_32x32x2_f32: direct 1x1/4x4/7x7 layers as implicit GEMM, the 3x3 layers as Winograd F(4x4,3x3)/F(2x2,3x3) component GEMMs; %d launches per network evaluation) + wino_input/wino_output transform kernels"
                          % int(round(prof["conv_launches"] / max(prof["net_evals"], 1))),
                "avg_launch_ms": (prof["conv_ms"] + prof["wino_ms"]) / max(prof["conv_launches"], 1),
                "flops_per_launch": prof["conv_flops"] / max(prof["conv_launches"], 1),
                "executed_TFLOPs_mfma_kernels_only": exe,
                "executed_flops_per_launch": prof["conv_exec_flops"] / max(prof["conv_launches"], 1),
                "kernel_time_share": {"conv": prof["conv_ms"] / prof["wall_ms"], "layernorm": prof["ln_ms"] / prof["wall_ms"],
                                      "attention": prof["attn_ms"] / prof["wall_ms"], "winograd_transforms": prof["wino_ms"] / prof["wall_ms"],
                                      "other": prof["other_ms"] / prof["wall_ms"]},
                # north_star also asks for the HBM-roofline fraction: ideal-fusion conv bytes / wall / 8 TB/s
                "hbm_algorithmic_GBps": prof["conv_bytes"] / (prof["wall_ms"] * 1e-3) / 1e9,
                "hbm_frac": prof["conv_bytes"] / (prof["wall_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                "whole_path_TFLOPs": prof["conv_flops"] / (prof["wall_ms"] * 1e-3) / 1e12,
            }
            if a.dtype != "fp32":
                # the bf16 kernel is fed from fp32 activations: HBM binds, so quote the HBM roof first
                r = res["roofline"]
                gbps = prof["conv_bytes"] / (prof["conv_ms"] * 1e-3) / 1e9
                r.update({"bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
                          "mfma_TFLOPs": alg, "mfma_frac": alg / PEAK_BF16_TFLOPS,
                          "kernel": "conv_igemm_kernel<bf16> (v_mfma_f32_32x32x16_bf16 implicit GEMM, fp32 activations rounded while staged; "
                                    "achieved = ideal-fusion conv bytes / conv kernel time)"})
        if not a.no_cpu_baseline and world == 1 and a.model == "unet":
            try:
                res["cpu_baseline"] = cpu_baseline(a.size, a.T)
            except Exception as ex:  # the GPU number must not be lost to a host-side problem
                res["cpu_baseline"] = {"value": None, "error": repr(ex)}
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
