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
`roofline.dominant_kernel` names the kernel class with the largest share of that pass and its own fraction of the roof.
`roofline.traffic` (r05): HBM bytes per conv launch MEASURED in the default 1-GPU run — after the timed region the script spawns two bounded rocprofv3
counter passes of itself (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, T = 3; `live_traffic()`); without rocprofv3 (or with `--no-live-pmc` / `--no-secondary`)
it falls back to the committed summary of the builder's passes; `traffic_source` says which.
`cpu_baseline`: the torch-CPU port of the reference timed on this box's host cores on a bounded sample.

`--scaling strong` splits ONE global batch of --batch images over the ranks (north_star: "image batches shard across the 8
GPUs"); the default, weak, gives every rank its own --batch images.  `rank_ms_per_step` lists min / max over the ranks.

`secondary` (default N-rank run only, after the headline is timed; `--no-secondary` skips it): the other BASELINE.json
configs and the 512x512 UNet batch, each timed for ONE sampler call on the same production path and carrying its own
roofline fraction from a 3-evaluation event pass.  They never change the headline keys.
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
            "sample": "B=1 %dx%d, %d of T=%d reverse_sde steps timed (%.2f s/step), extrapolated; torch %s CPU, %d threads of %d logical CPUs; "
                      "kind=port: oracle/torch_cpu_port.py (golden-checked against the reference, which does not exist on the GPU box; the "
                      "reference itself in the build container: 0.034 images/s at 1x128x128 T=100, 8 vCPU)"
                      % (size, size, n, T, per_step, torch.__version__, cores, os.cpu_count() or 0)}


# kernel classes of the event pass: description prefix (engine_plan.hip `op.desc`) -> the HIP kernel behind it (short names: the
# whole JSON line has to fit the driver's 8 KB record; what each kernel does is in DESIGN.md section 3)
KERNEL_CLASSES = [
    ("conv(winograd F4 fused)", "wino4_fused64p_kernel"),   # (r06: the class = wino4_fused64p_kernel + wino4_fused64t_kernel, whichever the plan picked per layer)
    ("conv(winograd", "gemm_zloop_kernel (Winograd component GEMMs)"),
    ("conv(split f16x2 winograd F4 fused)", "wino4_fused64p_kernel<PAIR>"),
    ("conv(split f16x2) M=", "conv_igemm_kernel<PAIR>"),
    ("conv(split bf16x2) M=", "conv_igemm_kernel<PAIR>"),
    ("conv(split", "gemm_split2i_kernel"),
    ("conv(bf16) M=", "conv_igemm_kernel<bf16>"),
    ("conv(fp16) M=", "conv_igemm_kernel<f16>"),
    ("conv M=", "conv_igemm_kernel / gemm_zloop_kernel (direct layers)"),
    ("conv", "conv kernels (other)"),
    ("naf_chain", "naf_chain_kernel"),
    ("wino_", "wino_input_kernel / wino_output_kernel"),
    ("linear_attention", "attn_kv_ctx_kernel / attn_q_out_fused_kernel"),
    ("full_attention", "full_attn_kernel"),
    ("layernorm", "layernorm_kernel"),
]
OTHER_CLASS = "other (prep / FiLM row / update / pointwise)"


def op_class(desc):
    name = next((n for pre, n in KERNEL_CLASSES if desc.startswith(pre)), OTHER_CLASS)
    if name in ("conv_igemm_kernel<bf16>", "conv_igemm_kernel<f16>") and " k=3x3 s=1 " in desc:
        return "conv3x3_halo_bf16_kernel" + ("<f16>" if "f16" in name else "")   # LDS-resident halo tile (conv_halo.hip)
    return name


def op_classes(op_text):
    """Parse irsde_op_profile's text (one line per launch group of ONE network evaluation: `ms  description`) into
    {class name: [ms, executed flops, launches]}."""
    out = {}
    for line in op_text.splitlines():
        parts = line.split(None, 2)
        if len(parts) < 3 or parts[1] != "ms":
            continue
        ms, desc = float(parts[0]), parts[2]
        fl = float(desc.split("flops=")[1].split()[0]) if "flops=" in desc else 0.0
        ex = float(desc.split("exec=")[1].split()[0]) if "exec=" in desc else fl
        acc = out.setdefault(op_class(desc), [0.0, 0.0, 0])
        acc[0] += ms
        acc[1] += ex
        acc[2] += 1
    return out


def _r(x, nd=4):
    """Round to nd significant digits (keeps the JSON line short)."""
    return float("%.*g" % (nd, x)) if isinstance(x, float) else x


def roofline_object(prof, op_text, w):
    """`roofline` for one workload from the event-instrumented pass (see module docstring)."""
    conv_t = (prof["conv_ms"] + prof["wino_ms"]) * 1e-3
    ach = prof["conv_exec_flops"] / conv_t / 1e12
    exe = prof["conv_exec_flops"] / (prof["conv_ms"] * 1e-3) / 1e12
    alg = prof["conv_flops"] / conv_t / 1e12
    fp32 = w["dtype"] in ("fp32", "fp32_split", "fp32_split_f16")
    peak = PEAK_FP32_TFLOPS if fp32 else PEAK_BF16_TFLOPS
    classes = op_classes(op_text)
    tot = sum(v[0] for v in classes.values()) or 1.0
    nev = max(prof["net_evals"], 1)
    r = {
        "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "mfma_kernel_frac": exe / peak,
        "algorithmic_equiv_TFLOPs": alg,
        "timing": "untimed pass after the timed region: hipEvent pair around every kernel on the engine stream, %d evaluations" % int(nev),
        "launches_per_evaluation": sum(v[2] for v in classes.values()),
        "conv_launches_per_evaluation": int(round(prof["conv_launches"] / nev)),
        "avg_launch_ms": (prof["conv_ms"] + prof["wino_ms"]) / max(prof["conv_launches"], 1),
        "executed_flops_per_launch": prof["conv_exec_flops"] / max(prof["conv_launches"], 1),
        "algorithmic_bytes_per_launch": prof["conv_bytes"] / max(prof["conv_launches"], 1),
        # north_star also asks for the HBM-roofline fraction: ideal-fusion conv bytes / wall / 8 TB/s
        "hbm_algorithmic_GBps": prof["conv_bytes"] / (prof["wall_ms"] * 1e-3) / 1e9,
        "hbm_frac": prof["conv_bytes"] / (prof["wall_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBPS,
        "whole_path_TFLOPs": prof["conv_flops"] / (prof["wall_ms"] * 1e-3) / 1e12,
    }

    def products(n):   # 16-bit MFMA products a split-mode kernel issues per f32 product (0: the kernel runs at `peak`)
        if w["dtype"] not in ("fp32_split", "fp32_split_f16"):
            return 0.0
        return 4.0 if n.startswith("wino4_fused64p_kernel<PAIR>") else 3.0 if (n.startswith("gemm_split2i") or "<PAIR>" in n) else 0.0

    def kfrac(n, v):
        tf = v[1] / (v[0] * 1e-3) / 1e12
        return products(n) * tf / PEAK_BF16_TFLOPS if products(n) else tf / peak

    # per kernel class of ONE evaluation, by time: share of the evaluation, ms, launches, executed TFLOP/s and fraction of its roof
    per = [{"name": n, "share": _r(v[0] / tot, 3), "ms": _r(v[0], 4), "n": v[2],
            **({"TFLOPs": _r(v[1] / (v[0] * 1e-3) / 1e12, 4), "frac": _r(kfrac(n, v), 3)} if v[1] > 0 and v[0] > 0 else {})}
           for n, v in sorted(classes.items(), key=lambda kv: -kv[1][0])]
    r["per_kernel"] = per
    dom = next((k for k in per if "frac" in k), None)
    if dom:
        r["dominant_kernel"] = {"name": dom["name"], "time_share": dom["share"], "ms_per_evaluation": dom["ms"],
                                "executed_TFLOPs": dom["TFLOPs"], "frac": dom["frac"]}
    if w["dtype"] in ("fp32_split", "fp32_split_f16"):
        # two pipes in one evaluation: `frac` = the pipe time the executed work needs at each kernel's own roof / the measured
        # MFMA-kernel time (a true fraction); `achieved` stays f32-equivalent (FLOPs counted once per f32 product)
        need = nev * sum(v[1] * (products(n) / (PEAK_BF16_TFLOPS * 1e12) if products(n) else 1.0 / (PEAK_FP32_TFLOPS * 1e12))
                         for n, v in classes.items())
        r["frac"] = need / conv_t
        r["mfma_kernel_frac"] = need / (prof["conv_ms"] * 1e-3)
        r["peak"] = None
        r["achieved_vs_f32_roof"] = ach / PEAK_FP32_TFLOPS
    if not fp32:
        # 16-bit operands: 16x the MFMA rate turns the convolutions L2/HBM-bound, so quote the HBM roof first
        gbps = prof["conv_bytes"] / (prof["conv_ms"] * 1e-3) / 1e9
        r.update({"bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
                  "mfma_TFLOPs": alg, "mfma_frac": alg / PEAK_BF16_TFLOPS})
    return {k: _r(v) for k, v in r.items()}


def live_traffic(timeout_s=150):
    """HBM bytes per conv launch of the headline workload, MEASURED in this run: two rocprofv3 counter passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`:
    separate runs, as guides/MI355X_MICROARCH.md prescribes; `--kernel-trace` only, no other trace domain) of THIS file on the production plan
    (`--steps 1 --warmup 0 --T 3`), summarised like tools/pmc_summary.py: raw unit KiB, read bytes = 2 x FETCH_SIZE on gfx950, conv + Winograd-transform
    kernels / conv launches.  Each pass is a bounded subprocess; any failure (no rocprofv3, timeout, unexpected CSV) returns None and the caller
    falls back to the committed summary."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe or os.environ.get("IRSDE_BENCH_NO_LIVE_PMC") == "1":
        return None
    tmp = tempfile.mkdtemp(prefix="irsde_pmc_", dir="/tmp")
    conv_names = ("conv_igemm", "gemm_zloop", "conv3x3_halo", "wino4_fused", "conv3x3_narrow", "gemm_split")
    try:
        kib, launches = {}, 0
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", os.path.join(tmp, ctr), "-o", "pmc", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--T", "3", "--no-profile", "--no-cpu-baseline",
                   "--no-secondary", "--no-live-pmc"]
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=timeout_s, check=True)
            files = glob.glob(os.path.join(tmp, ctr, "**", "*counter_collection.csv"), recursive=True)
            conv, wino, n = 0.0, 0.0, 0
            for r in csv.DictReader(open(files[0])):
                if r.get("Counter_Name", ctr) != ctr:
                    continue
                name = r["Kernel_Name"]
                if any(k in name for k in conv_names):
                    conv += float(r["Counter_Value"])
                    n += 1
                elif "wino_" in name:
                    wino += float(r["Counter_Value"])
            kib[ctr] = (conv, wino)
            launches = n
        if launches <= 0:
            return None
        conv_b = (2 * kib["FETCH_SIZE"][0] + kib["WRITE_SIZE"][0]) * 1024
        wino_b = (2 * kib["FETCH_SIZE"][1] + kib["WRITE_SIZE"][1]) * 1024
        return {"traffic": (conv_b + wino_b) / launches, "conv_only": conv_b / launches, "launches": launches}
    except Exception:  # noqa: BLE001 - measurement extra: never lose the headline over it
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


DTYPE_LABEL = {"fp32": "f32", "bf16": "bf16 operands / f32 accumulate+state",
               "bf16_act": "bf16 operands + bf16 activation storage / f32 accumulate+state",
               "fp16": "f16 operands / f32 accumulate+state",
               "fp32_split": "f32 storage+state; GEMM operands as bf16 hi+lo pairs (3 products on the bf16 MFMA)",
               "fp32_split_f16": "f32 storage+state; GEMM operands as fp16 hi+lo pairs (22+ significand bits, 3 products on the f16 MFMA)"}

# the other BASELINE.json configs + the 512x512 batch north_star names: timed after the headline, reported under `secondary`
# (tags are short on purpose: every entry has to survive in the driver's 8 KB record; DESIGN.md section 5 spells them out)
SECONDARY = [
    dict(tag="configs[2] unet 16x256 ode bf16_act", model="unet", dtype="bf16_act", mode="ode", batch=16, size=256, T=100),
    dict(tag="configs[3] nafnet 8x512 T=200 f32", model="nafnet", dtype="fp32", mode="sde", batch=8, size=512, T=200),
    dict(tag="configs[4] latent 64x(64x64x4) fp16", model="latent", dtype="fp16", mode="sde", batch=64, size=256, T=100),
    dict(tag="unet 16x512 f32", model="unet", dtype="fp32", mode="sde", batch=16, size=512, T=100),
    dict(tag="configs[1] in fp32_split_f16", model="unet", dtype="fp32_split_f16", mode="sde", batch=16, size=256, T=100),
    dict(tag="configs[3] in fp32_split_f16", model="nafnet", dtype="fp32_split_f16", mode="sde", batch=8, size=512, T=200),
    dict(tag="strong-scaling shard: unet 2x256 f32", model="unet", dtype="fp32", mode="sde", batch=2, size=256, T=100),
    # the per-GPU shard of "batch=64 ... 8xMI355X" (BASELINE configs[4]): 8 images per GPU
    dict(tag="configs[4] shard of 8 GPUs: latent 8x(64x64x4) fp16", model="latent", dtype="fp16", mode="sde", batch=8, size=256, T=100),
]


def workload_text(w, n_evals):
    m, fmt = w["model"], (w["mode"], w["batch"], w["size"], w["size"], w["T"])
    if m == "dsde":
        return ("denoising-sde unconditional UNet nf=64 depth=4 (full attention at the bottleneck), DenoisingSDE reverse_%s from the optimal "
                "timestep of sigma=25 (" + str(n_evals) + " network evaluations), batch=%d/GPU %dx%d, schedule T=%d") % fmt
    if m == "latent":
        return ("Latent-Refusion (latent-bokeh): latent UNet ch=64 [1,2,4] embed 4 (encode + decode once per image) + lens-conditioned "
                "ConditionalNAFNet width=64 enc[1,1,1,28] on the 64x64x4 latent, reverse_%s, batch=%d/GPU %dx%d, T=%d (BASELINE.json configs[4] shape)") % fmt
    if m == "nafnet":
        return "Refusion ConditionalNAFNet width=64 enc[1,1,1,28], reverse_%s, batch=%d/GPU %dx%d, T=%d (BASELINE.json configs[3] network)" % fmt
    return "IR-SDE deraining ConditionalUNet nf=64 depth=4, reverse_%s, batch=%d/GPU %dx%d, T=%d (BASELINE.json configs[1])" % fmt


DIST_ON = False   # a process group is up (world > 1, or the one-rank RCCL test hook): barriers, gathers and all_reduces are live


class Workload:
    """One synthetic workload on this rank: models, schedule, resident inputs, `one_step()` = one pass of the hot path over
    this rank's share of the batch (+ the final gather when world > 1), `profile_pass(T)` = the event-instrumented pass."""

    def __init__(self, P, w, dev, rank, world, scaling):
        import numpy as np
        import torch
        self.P, self.w, self.rank, self.world = P, w, rank, world
        model_kind = w["model"]
        self.latent_model = None
        if model_kind == "latent":  # latent-bokeh/options/bokeh/test/refusion.yml: UNet ch 64 [1,2,4] embed 4 (256^2 -> 64x64x4) + NAFNet
            ncfg = dict(width=64, enc_blk_nums=[1, 1, 1, 28], middle_blk_num=1, dec_blk_nums=[1, 1, 1, 1])
            model = P.latent_bokeh.ConditionalNAFNet(img_channel=4, **ncfg)   # lens-conditioned (lens_info kwargs)
            lm = P.latent.UNet(in_ch=3, out_ch=3, ch=64, ch_mult=[1, 2, 4], embed_dim=4)
            lm.load_state_dict(synth_state_dict(lm, 1))
            if w["dtype"] == "fp16":   # configs[4] names fp16: the encode / decode convolutions run on fp16 operands too (tests/test_gpu_fullres.py::test_latent_pipeline_256_all_fp16)
                lm.engine_flags = P._lib.FLAG_FP16
            self.latent_model = lm.to(dev).eval()
        elif model_kind == "dsde":  # denoising-sde/options/test/ir-sde.yml: unconditional UNet + DenoisingSDE(max_sigma=75, T=100)
            model = P.denoising_sde.ConditionalUNet(3, 3, 64, depth=4)
        elif model_kind == "nafnet":  # refusion.yml network_G
            ncfg = dict(width=64, enc_blk_nums=[1, 1, 1, 28], middle_blk_num=1, dec_blk_nums=[1, 1, 1, 1])
            model = P.ConditionalNAFNet(img_channel=3, **ncfg)
        else:
            model = P.ConditionalUNet(3, 3, 64, depth=4)
        model.load_state_dict(synth_state_dict(model, 0))
        model = model.to(dev).eval()
        model.set_compute_dtype(w["dtype"])
        model.engine_flags |= int(w.get("extra_flags") or 0)   # --engine-flags: measurement only (e.g. 8192 = IRSDE_FLAG_NO_NAF_CHAIN, 4096 = IRSDE_FLAG_NO_FUSED_ATTN)
        self.model = model
        max_sigma = w.get("max_sigma")
        max_sigma = max_sigma if max_sigma is not None else {"nafnet": 50, "dsde": 75, "latent": 50}.get(model_kind, 10)
        T = w["T"]
        if model_kind == "dsde":
            sde = P.DenoisingSDE(max_sigma=max_sigma, T=T, device=dev)
        else:
            sde = P.IRSDE(max_sigma=max_sigma, T=T, schedule="cosine", eps=0.005, device=dev)
        sde.set_model(model)
        sde.seed = 7
        sde.profile = False    # the timed region runs the production path: hipGraph replay, no per-kernel events
        sde.use_graph = True
        self.sde = sde
        # weak: every rank samples its own `batch` images; strong: ONE global batch of `batch` images split over the ranks
        if scaling == "strong":
            self.nglobal = w["batch"]
            self.lo, self.hi = P.shard_bounds(self.nglobal, world, rank)
        else:
            self.nglobal = w["batch"] * world
            self.lo, self.hi = rank * w["batch"], (rank + 1) * w["batch"]
        nloc = self.hi - self.lo
        # every rank materialises only its shard of the synthetic global batch
        lq, xT = synth_inputs(1234, max(nloc, 1), w["size"], w["size"], max_sigma)
        rs = np.random.RandomState(1000 + rank)
        lq = np.clip(lq + 0.01 * rs.standard_normal(lq.shape).astype(np.float32), 0, 1)
        self.mu = torch.from_numpy(lq[:nloc]).to(dev)
        self.x_T = torch.from_numpy(xT[:nloc]).to(dev)
        self.n_evals = T
        self.fn = None
        if model_kind == "dsde":  # denoising-sde/test.py:103-107: reverse_ode from the optimal timestep of the noise level (sigma 25)
            self.n_evals = n = int(sde.get_optimal_timestep(25))
            self.fn = (lambda x: sde.reverse_ode(x, T=n)) if w["mode"] != "sde" else (lambda x: sde.reverse_sde(x, T=n))
        elif model_kind == "latent":  # latent-dehazing/test.py:90-100: encode once, sample in the latent, decode once
            sample = {"sde": sde.reverse_sde, "ode": sde.reverse_ode, "posterior": sde.reverse_posterior}[w["mode"]]
            lens_rs = np.random.RandomState(5 + rank)   # per image: src_lens, tgt_lens, disparity (latent-bokeh/test.py:91)
            self.lens_info = [torch.from_numpy(lens_rs.uniform(a, b, max(nloc, 1)).astype(np.float32)[:nloc]) for a, b in ((1.4, 2.8), (1.8, 16.0), (0.0, 100.0))]
            lm, mu = self.latent_model, self.mu

            def fn(_unused, T=-1):
                latent_LQ, hidden = lm.encode(mu)
                sde.set_mu(latent_LQ)
                return lm.decode(sample(sde.noise_state(latent_LQ), T=T, lens_info=self.lens_info), hidden)
            self.fn = fn
        else:
            sde.set_mu(self.mu)

    def one_step(self):
        P, sde = self.P, self.sde
        if self.fn is not None:
            out = self.fn(self.x_T)
            return P.gather_batch(out, self.nglobal) if DIST_ON else out
        # the product's N>1 path: sample this rank's shard, then the final all_gather (the only collective)
        sde.image_offset = 0
        return P.sample_shard(sde, self.w["mode"], self.x_T, self.mu, self.lo, self.nglobal)

    def short_call(self, T):
        """The same call with only the last T steps (plan building, graph capture, the event pass of the secondaries)."""
        sde = self.sde
        if self.w["model"] == "latent":
            return self.fn(self.x_T, T=T)
        if self.fn is not None:
            return self.fn(self.x_T)
        sde.image_offset = self.lo
        sde.set_mu(self.mu)
        return {"sde": sde.reverse_sde, "ode": sde.reverse_ode, "posterior": sde.reverse_posterior}[self.w["mode"]](self.x_T, T=T)

    def profile_pass(self, T):
        """Untimed: the sampling call once more with a hipEvent pair around every kernel (eager launches)."""
        import ctypes
        import torch
        from image_restoration_sde_amd import _lib
        sde = self.sde
        sde.profile = True
        try:
            self.short_call(T)
            torch.cuda.synchronize()
            prof = sde.last_profile()
            buf = ctypes.create_string_buffer(1 << 18)
            _lib.check(_lib.lib().irsde_op_profile(self.model.engine().h, buf, len(buf)))
        finally:
            sde.profile = False
        return prof, buf.value.decode()


def timed(wl, steps, warmup, world, dev):
    """W untimed + K timed passes bracketed by barrier + synchronize; returns (max over ranks, [per-rank seconds], last output)."""
    import torch
    import torch.distributed as dist

    def fence():
        torch.cuda.synchronize()
        if DIST_ON:
            dist.barrier()
        torch.cuda.synchronize()

    out = None
    for _ in range(warmup):
        out = wl.one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = wl.one_step()
    torch.cuda.synchronize()
    mine = time.perf_counter() - t0      # this rank's own finish time (before waiting for the others)
    fence()
    dt = time.perf_counter() - t0
    per_rank = [mine]
    if DIST_ON:
        tt = torch.tensor([dt, mine], dtype=torch.float64, device=dev if dist.get_backend() != "gloo" else "cpu")
        parts = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(parts, tt)
        dt = max(float(p[0]) for p in parts)
        per_rank = [float(p[1]) for p in parts]
    return dt, per_rank, out


def main():
    # RCCL / device-tensor sharing across the ranks of one node needs dmabuf IPC on this driver; the GPU box exports this already — set it for any other launcher too
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU (weak) or in the global batch (strong); BASELINE configs[1]: 16")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--T", type=int, default=100)
    ap.add_argument("--mode", default="sde", choices=["sde", "ode", "posterior"])
    ap.add_argument("--model", default="unet", choices=["unet", "nafnet", "dsde", "latent"],
                    help="unet: IR-SDE ConditionalUNet (BASELINE configs[1]); nafnet: Refusion ConditionalNAFNet (configs[3])")
    ap.add_argument("--max-sigma", type=float, default=None)
    ap.add_argument("--dtype", default="fp32", choices=sorted(DTYPE_LABEL),
                    help="bf16 = BASELINE configs[2] (conv operands bf16, fp32 accumulate); fp16 = configs[4] (IEEE fp16 operands); "
                         "the headline metric is fp32")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch images per GPU; strong: one global batch of --batch images split over the GPUs")
    ap.add_argument("--engine-flags", type=int, default=0, help="measurement only: extra IRSDE_FLAG_* bits for the score network (A/B of plan choices); never part of a reported line's default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the untimed event-instrumented pass (no `roofline` object)")
    ap.add_argument("--no-secondary", action="store_true", help="headline workload only")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not spawn the two rocprofv3 counter passes that measure `roofline.traffic` (default N=1 run only)")
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
    # IRSDE_BENCH_FORCE_RCCL=1 (tests/test_gpu_fullres.py, 1-GPU box): a process group of ONE rank over RCCL, so that the N > 1 code path —
    # init_process_group("nccl", device_id=...), barrier, the device-side all_gather of gather_batch, the all_reduce of all_ok — runs on the hardware there is
    global DIST_ON
    force_rccl = world == 1 and os.environ.get("IRSDE_BENCH_FORCE_RCCL") == "1"
    if force_rccl:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    DIST_ON = world > 1 or force_rccl
    if DIST_ON:
        if oversub:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    if a.scaling == "strong" and a.batch < world:
        sys.exit("bench.py: --scaling strong needs --batch >= --gpus (%d images over %d ranks leaves empty shards)" % (a.batch, world))

    def all_ok(ok):
        """True only if EVERY rank says ok (so a failure on one rank makes all ranks skip the phase together instead of leaving the
        others blocked in the next collective)."""
        if not DIST_ON:
            return ok
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    head = dict(model=a.model, dtype=a.dtype, mode=a.mode, batch=a.batch, size=a.size, T=a.T, max_sigma=a.max_sigma, extra_flags=a.engine_flags)
    is_default = (a.model, a.dtype, a.mode, a.batch, a.size, a.T, a.max_sigma, a.scaling, a.engine_flags) == ("unet", "fp32", "sde", 16, 256, 100, None, "weak", 0)
    t_setup = time.perf_counter()
    wl = Workload(P, head, dev, rank, world, a.scaling)
    wl.short_call(2)          # untimed: weight upload + repack, plan build, graph capture (what 8 simultaneous ranks would contend on)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup
    setup_s = [t_setup]
    if DIST_ON:
        parts = [None] * world
        dist.all_gather_object(parts, t_setup)
        setup_s = [float(x) for x in parts]
    dt, per_rank, out = timed(wl, a.steps, a.warmup, world, dev)
    assert out.shape[0] == wl.nglobal and bool(torch.isfinite(out).all())

    prof = None
    if not a.no_profile and rank == 0 and a.model != "dsde" and wl.hi > wl.lo:
        prof, op_text = wl.profile_pass(a.T)

    gather_txt = "gloo, ranks sharing GPUs: TEST HOOK, not a scaling number" if (world > 1 and oversub) else "RCCL"
    res = None
    if rank == 0:
        imgs = wl.nglobal * a.steps
        res = {
            "metric": ("restored images/sec at %dx%d, %d-step DenoisingSDE reverse sampler (optimal timestep of sigma=25)" % (a.size, a.size, wl.n_evals)
                       if a.model == "dsde" else "restored images/sec at %dx%d, %d-step IR-SDE reverse sampler" % (a.size, a.size, a.T)),
            "value": imgs / dt, "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1000.0 * dt / a.steps, "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": None, "dtype": DTYPE_LABEL[a.dtype], "data": "synthetic",
            "config": {"workload": workload_text(head, wl.n_evals), "compute_dtype": a.dtype, "global_batch": wl.nglobal,
                       "parallelism": "batch-shard x%d (%s scaling; no collective in the T loop; final all_gather over %s)" % (world, a.scaling, gather_txt)},
            "rank_ms_per_step": {"min": _r(1000.0 * min(per_rank) / a.steps), "max": _r(1000.0 * max(per_rank) / a.steps)},
            # per rank: model build + weight upload / repack + plan + graph capture + a 2-step sampler call, before the timed region
            "rank_setup_s": {"min": _r(min(setup_s), 3), "max": _r(max(setup_s), 3)},
        }
        if a.engine_flags:
            res["config"]["engine_flags_measurement_only"] = a.engine_flags
        if prof is not None and prof["conv_ms"] > 0:
            r = roofline_object(prof, op_text, head)
            # HBM bytes per conv launch: rocprofv3 PMC passes cannot run inside this process; the figure comes from the
            # builder's PMC run of this same command, committed under profiles/ (source named next to it)
            traffic, traffic_src = None, None
            live = live_traffic() if (is_default and world == 1 and not a.no_live_pmc and not a.no_secondary) else None   # (the full default run only: tools/ run this file UNDER rocprofv3 with --no-secondary)
            if live:
                traffic = live["traffic"]
                traffic_src = ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate subprocess passes of "
                               "bench.py --steps 1 --warmup 0 --T 3), %d conv launches, conv kernels alone %.0f bytes per launch" % (live["launches"], live["conv_only"]))
            try:
                if is_default and traffic is None:
                    import re as _re   # the headline workload's file only: r<NN>_final_bench_pmc_hbm.json (not the latent / NAFNet / split-mode ones)
                    pm = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if _re.fullmatch(r"r\d+_final_bench_pmc_hbm\.json", f))
                    traffic = json.load(open(os.path.join(ROOT, "profiles", pm[-1])))["traffic_bytes_per_launch"]
                    traffic_src = "profiles/%s (builder's rocprofv3 PMC passes of this command; not measured in this run)" % pm[-1]
            except (OSError, IndexError, KeyError, ValueError):
                traffic, traffic_src = None, None
            r.update({"traffic": traffic, "traffic_source": traffic_src,
                      "traffic_unit": "bytes per conv launch (2 x FETCH_SIZE + WRITE_SIZE)",
                      # the same figure per network evaluation (x the conv launches of one evaluation) next to the algorithmic bytes
                      "traffic_per_evaluation": (traffic * r["conv_launches_per_evaluation"]) if traffic else None,
                      "algorithmic_bytes_per_evaluation": r["algorithmic_bytes_per_launch"] * r["conv_launches_per_evaluation"]})
            res["roofline"] = r

    # ---- secondary workloads: the other BASELINE configs, after the headline has been timed.  Every phase that ends in a collective
    # is entered only if ALL ranks got there (all_ok), so one rank's failure (OOM, plan error) cannot strand the others and lose the headline
    if is_default and not a.no_secondary and not oversub:
        del wl, out
        torch.cuda.empty_cache()
        sec = []
        for w in SECONDARY:
            entry = {"tag": w["tag"]}
            swl, ok = None, True
            try:
                swl = Workload(P, w, dev, rank, world, "weak")
                swl.short_call(3)                                   # untimed: plan, graph capture, clocks
            except Exception as ex:  # noqa: BLE001
                entry["error"], ok = repr(ex)[:200], False
            if all_ok(ok):
                try:
                    sdt, sranks, sout = timed(swl, 1, 0, world, dev)
                    assert sout.shape[0] == swl.nglobal and bool(torch.isfinite(sout).all())
                    entry.update({"value": _r(swl.nglobal / sdt), "unit": "images/s", "ms_per_step": _r(1000.0 * sdt), "steps": 1,
                                  "ms_per_evaluation": _r(1000.0 * sdt / swl.n_evals), "n_gpus": world, "global_batch": swl.nglobal,
                                  "mode": "reverse_" + w["mode"], "T": w["T"], "dtype": w["dtype"]})
                    if rank == 0:
                        sp, sop = swl.profile_pass(3)
                        if sp["conv_ms"] > 0:
                            rr = roofline_object(sp, sop, w)
                            keep = {k: rr[k] for k in ("bound", "achieved", "unit", "frac", "mfma_kernel_frac", "hbm_frac", "whole_path_TFLOPs",
                                                       "launches_per_evaluation", "achieved_vs_f32_roof") if k in rr}
                            if "dominant_kernel" in rr:
                                keep["dominant_kernel"] = {k: rr["dominant_kernel"][k] for k in ("name", "time_share", "frac")}
                            entry["roofline"] = keep
                    del sout
                except Exception as ex:  # noqa: BLE001
                    entry["error"] = repr(ex)[:200]
            elif "error" not in entry:
                entry["error"] = "skipped: another rank failed to set this workload up"
            del swl
            torch.cuda.empty_cache()
            sec.append(entry)
        if rank == 0:
            res["secondary"] = sec

    if rank == 0:
        if not a.no_cpu_baseline and world == 1 and a.model == "unet":
            try:
                res["cpu_baseline"] = cpu_baseline(a.size, a.T)
            except Exception as ex:  # the GPU number must not be lost to a host-side problem
                res["cpu_baseline"] = {"value": None, "error": repr(ex)}
        print(json.dumps(res, separators=(",", ":")))
    if DIST_ON:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
