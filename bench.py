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
            "sample": "B=1 %dx%d, %d of T=%d reverse_sde steps timed (%.2f s/step), extrapolated x%d; torch %s CPU, "
                      "%d threads (affinity/cgroup quota; host has %d logical CPUs). kind=port: oracle/torch_cpu_port.py, the "
                      "reference's algorithm restated in torch-CPU functional ops and golden-checked against the reference "
                      "(tests/test_oracle_golden.py) - /root/reference itself does not exist on the GPU box.  The unmodified "
                      "reference classes measured in the build container (SURVEY.md 8d): 0.034 images/s at 1x128x128, T=100, 8 vCPU"
                      % (size, size, n, T, per_step, T, torch.__version__, cores, os.cpu_count() or 0)}


# kernel classes of the event pass: description prefix (engine_plan.hip `op.desc`) -> the HIP kernels behind it
KERNEL_CLASSES = [
    ("conv(winograd F4 fused)", "wino4_fused64_kernel / wino4_fused_kernel (Winograd F(4x4,3x3): input transform + 36 component GEMMs + output transform + epilogue in one kernel)"),
    ("conv(winograd", "gemm_zloop_kernel (component GEMMs of the three-launch Winograd layers)"),
    ("conv(split f16x2 winograd F4 fused)", "wino4_fused64_kernel<PAIR> (the fused Winograd kernel with f32 operands as fp16 hi+lo pairs: 4 cross products on v_mfma_f32_16x16x32_f16; FLOPs counted once per f32 product)"),
    ("conv(split", "gemm_split2i_kernel (f32 operands as 16-bit hi+lo pairs, 3 cross products on v_mfma_f32_32x32x16_bf16 / _f16; FLOPs counted once per f32 product)"),
    ("conv M=", "conv_igemm_kernel / gemm_zloop_kernel (direct implicit-GEMM layers: 1x1, 4x4 s2, 7x7, narrow 3x3)"),
    ("conv", "conv kernels (other)"),
    ("wino_", "wino_input_kernel / wino_output_kernel (transforms of the three-launch Winograd layers)"),
    ("linear_attention", "attn_kv_ctx_kernel / attn_q_out_fused_kernel / attn_ctx_* (LinearAttention)"),
    ("full_attention", "full_attn_kernel"),
    ("layernorm", "layernorm_kernel"),
]


def op_classes(op_text):
    """Parse irsde_op_profile's text (one line per launch group of ONE network evaluation: `ms  description`) into
    {class name: [ms, executed flops]}."""
    out = {}
    for line in op_text.splitlines():
        parts = line.split(None, 2)
        if len(parts) < 3 or parts[1] != "ms":
            continue
        ms, desc = float(parts[0]), parts[2]
        name = next((n for pre, n in KERNEL_CLASSES if desc.startswith(pre)), "other (prep / FiLM row / state update / NAFNet pointwise)")
        fl = float(desc.split("flops=")[1].split()[0]) if "flops=" in desc else 0.0
        ex = float(desc.split("exec=")[1].split()[0]) if "exec=" in desc else fl
        acc = out.setdefault(name, [0.0, 0.0])
        acc[0] += ms
        acc[1] += ex
    return out


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
    mfma_classes = sorted(((n, v) for n, v in classes.items() if v[1] > 0), key=lambda kv: -kv[1][0])
    r = {
        "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "mfma_kernel_frac": exe / peak,
        "algorithmic_equiv_TFLOPs": alg,
        "timing": "separate untimed pass after the timed region: eager launches, hipEvent pair around every kernel on the engine stream, %d network evaluations" % int(prof["net_evals"]),
        "kernel": "; ".join("%s: %.1f%% of the evaluation" % (n, 100.0 * v[0] / tot) for n, v in sorted(classes.items(), key=lambda kv: -kv[1][0])),
        "launches_per_evaluation": int(round(prof["conv_launches"] / max(prof["net_evals"], 1))),
        "avg_launch_ms": (prof["conv_ms"] + prof["wino_ms"]) / max(prof["conv_launches"], 1),
        "flops_per_launch": prof["conv_flops"] / max(prof["conv_launches"], 1),
        "executed_flops_per_launch": prof["conv_exec_flops"] / max(prof["conv_launches"], 1),
        "algorithmic_bytes_per_launch": prof["conv_bytes"] / max(prof["conv_launches"], 1),
        "executed_TFLOPs_mfma_kernels_only": exe,
        "kernel_time_share": {"conv": prof["conv_ms"] / prof["wall_ms"], "layernorm": prof["ln_ms"] / prof["wall_ms"],
                              "attention": prof["attn_ms"] / prof["wall_ms"], "winograd_transforms": prof["wino_ms"] / prof["wall_ms"],
                              "other": prof["other_ms"] / prof["wall_ms"]},
        # north_star also asks for the HBM-roofline fraction: ideal-fusion conv bytes / wall / 8 TB/s
        "hbm_algorithmic_GBps": prof["conv_bytes"] / (prof["wall_ms"] * 1e-3) / 1e9,
        "hbm_frac": prof["conv_bytes"] / (prof["wall_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBPS,
        "whole_path_TFLOPs": prof["conv_flops"] / (prof["wall_ms"] * 1e-3) / 1e12,
    }
    if mfma_classes:
        n, v = mfma_classes[0]
        r["dominant_kernel"] = {"name": n, "time_share": v[0] / tot, "ms_per_evaluation": v[0],
                                "executed_TFLOPs": v[1] / (v[0] * 1e-3) / 1e12, "frac": v[1] / (v[0] * 1e-3) / 1e12 / peak}
        r["per_kernel"] = [{"name": n, "time_share": v[0] / tot, "ms_per_evaluation": v[0], "executed_TFLOPs": v[1] / (v[0] * 1e-3) / 1e12,
                            "frac": v[1] / (v[0] * 1e-3) / 1e12 / peak} for n, v in mfma_classes]
    if w["dtype"] in ("fp32_split", "fp32_split_f16"):
        # two pipes in one evaluation: a pair kernel issues 3 16-bit MFMA products per f32 product.  `frac` = the pipe time the executed
        # work needs at each kernel's own roof / the measured MFMA-kernel time: a true fraction (<= 1); `achieved` stays f32-equivalent
        def products(n):   # 16-bit MFMA products per f32 product (0: an f32-MFMA kernel)
            return 3.0 if n.startswith("gemm_split2i_kernel") else 4.0 if n.startswith("wino4_fused64_kernel<PAIR>") else 0.0
        # (`classes` is ONE evaluation of the plan, the profile's times cover net_evals of them)
        need = max(prof["net_evals"], 1) * sum(v[1] * (products(n) / (PEAK_BF16_TFLOPS * 1e12) if products(n) else 1.0 / (PEAK_FP32_TFLOPS * 1e12))
                                               for n, v in classes.items())
        r["frac"] = need / conv_t
        r["mfma_kernel_frac"] = need / (prof["conv_ms"] * 1e-3)
        r["peak"] = None
        r["achieved_vs_f32_roof"] = ach / PEAK_FP32_TFLOPS
        for k in r.get("per_kernel", []) + ([r["dominant_kernel"]] if "dominant_kernel" in r else []):
            if products(k["name"]):
                k["frac"] = products(k["name"]) * k["executed_TFLOPs"] / PEAK_BF16_TFLOPS
        r["note"] = ("split modes: FLOPs are counted once per f32 product and compared with the f32 MFMA roof as a speed reference; the pair GEMMs "
                     "themselves execute 3 (the fused Winograd twin: 4) 16-bit MFMA products per f32 product on the bf16 / f16 pipe; frac = pipe time needed at each kernel's own roof / measured time")
    if not fp32:
        # 16-bit operands: 16x the MFMA rate turns the convolutions L2/HBM-bound, so quote the HBM roof first
        gbps = prof["conv_bytes"] / (prof["conv_ms"] * 1e-3) / 1e9
        r.update({"bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
                  "mfma_TFLOPs": alg, "mfma_frac": alg / PEAK_BF16_TFLOPS,
                  "achieved_note": "ideal-fusion conv bytes / conv kernel time (v_mfma_f32_32x32x16_bf16|f16 kernels)"})
    return r


DTYPE_LABEL = {"fp32": "f32", "bf16": "bf16 operands / f32 accumulate+state",
               "bf16_act": "bf16 operands + bf16 activation storage / f32 accumulate+state",
               "fp16": "f16 operands / f32 accumulate+state",
               "fp32_split": "f32 storage / state / transforms; deep Winograd component GEMMs and direct layers on bf16 hi+lo operand pairs (3 cross products on the bf16 MFMA, f32 accumulate)",
               "fp32_split_f16": "f32 storage / state / transforms; deep Winograd component GEMMs, direct layers and attention projections on fp16 hi+lo operand pairs (22+ significand bits: fp32-equivalent per layer; 3 cross products on the f16 MFMA, f32 accumulate)"}

# the other BASELINE.json configs + the 512x512 batch north_star names: timed after the headline, reported under `secondary`
SECONDARY = [
    dict(tag="BASELINE configs[2]: reverse_ode, bf16", model="unet", dtype="bf16_act", mode="ode", batch=16, size=256, T=100),
    dict(tag="BASELINE configs[3]: Refusion NAFNet 8x512x512 T=200", model="nafnet", dtype="fp32", mode="sde", batch=8, size=512, T=200),
    dict(tag="BASELINE configs[4]: Latent-Refusion 64x64x4 latent, batch 64, fp16", model="latent", dtype="fp16", mode="sde", batch=64, size=256, T=100),
    dict(tag="north_star 512x512 batch: IR-SDE UNet 16x512x512", model="unet", dtype="fp32", mode="sde", batch=16, size=512, T=100),
    dict(tag="BASELINE configs[1] workload in the opt-in fp32_split_f16 mode (IRSDE_FLAG_SPLIT_F16X2: fp32-equivalent per layer; error table: profiles/r03_split_error_table.txt)",
         model="unet", dtype="fp32_split_f16", mode="sde", batch=16, size=256, T=100),
    dict(tag="BASELINE configs[1] workload in the opt-in fp32_split mode (IRSDE_FLAG_SPLIT_BF16X2: 16-bit operand pairs, f32 exponent range)",
         model="unet", dtype="fp32_split", mode="sde", batch=16, size=256, T=100),
    dict(tag="BASELINE configs[3] workload (Refusion NAFNet 8x512x512 T=200) in the opt-in fp32_split_f16 mode",
         model="nafnet", dtype="fp32_split_f16", mode="sde", batch=8, size=512, T=200),
    dict(tag="north_star 512x512 batch (IR-SDE UNet 16x512x512) in the opt-in fp32_split_f16 mode",
         model="unet", dtype="fp32_split_f16", mode="sde", batch=16, size=512, T=100),
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
            return P.gather_batch(out, self.nglobal) if self.world > 1 else out
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
        if world > 1:
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
    if world > 1:
        tt = torch.tensor([dt, mine], dtype=torch.float64, device=dev if dist.get_backend() != "gloo" else "cpu")
        parts = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(parts, tt)
        dt = max(float(p[0]) for p in parts)
        per_rank = [float(p[1]) for p in parts]
    return dt, per_rank, out


def main():
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the untimed event-instrumented pass (no `roofline` object)")
    ap.add_argument("--no-secondary", action="store_true", help="headline workload only")
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
    if world > 1:
        if oversub:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    head = dict(model=a.model, dtype=a.dtype, mode=a.mode, batch=a.batch, size=a.size, T=a.T, max_sigma=a.max_sigma)
    is_default = (a.model, a.dtype, a.mode, a.batch, a.size, a.T, a.max_sigma, a.scaling) == ("unet", "fp32", "sde", 16, 256, 100, None, "weak")
    wl = Workload(P, head, dev, rank, world, a.scaling)
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
            "rank_ms_per_step": {"min": 1000.0 * min(per_rank) / a.steps, "max": 1000.0 * max(per_rank) / a.steps,
                                 "note": "each rank's own sampler time (before the closing barrier)"},
        }
        if prof is not None and prof["conv_ms"] > 0:
            r = roofline_object(prof, op_text, head)
            # HBM bytes per conv launch: rocprofv3 PMC passes cannot run inside this process; the figure comes from the
            # builder's PMC run of this same command, committed under profiles/ (source named next to it)
            traffic, traffic_src = None, None
            try:
                if is_default:
                    pm = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_bench_pmc_hbm.json"))
                    traffic = json.load(open(os.path.join(ROOT, "profiles", pm[-1])))["traffic_bytes_per_launch"]
                    traffic_src = "profiles/%s (builder's rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; not measured in this run)" % pm[-1]
            except (OSError, IndexError, KeyError, ValueError):
                traffic, traffic_src = None, None
            r.update({"traffic": traffic, "traffic_source": traffic_src,
                      "traffic_unit": "bytes per conv launch (2 x FETCH_SIZE + WRITE_SIZE, rocprofv3 PMC passes, profiles/*_bench_pmc_hbm.txt)"})
            res["roofline"] = r

    # ---- secondary workloads: the other BASELINE configs, after the headline has been timed; failures never lose the headline
    if is_default and not a.no_secondary and not oversub:
        del wl, out
        torch.cuda.empty_cache()
        sec = []
        for w in SECONDARY:
            entry = {"workload": None, "tag": w["tag"]}
            try:
                swl = Workload(P, w, dev, rank, world, "weak")
                entry["workload"] = workload_text(w, swl.n_evals)
                swl.short_call(3)                                   # untimed: plan, graph capture, clocks
                sdt, sranks, sout = timed(swl, 1, 0, world, dev)
                assert sout.shape[0] == swl.nglobal and bool(torch.isfinite(sout).all())
                entry.update({"value": swl.nglobal / sdt, "unit": "images/s", "ms_per_step": 1000.0 * sdt, "steps": 1,
                              "warmup": "one 3-step sampler call (plan + graph capture)", "n_gpus": world, "global_batch": swl.nglobal,
                              "dtype": DTYPE_LABEL[w["dtype"]], "mode": "reverse_" + w["mode"], "T": w["T"]})
                if rank == 0:
                    sp, sop = swl.profile_pass(3)
                    if sp["conv_ms"] > 0:
                        rr = roofline_object(sp, sop, w)
                        entry["roofline"] = {k: rr[k] for k in ("bound", "achieved", "peak", "unit", "frac", "mfma_kernel_frac", "hbm_frac",
                                                                "whole_path_TFLOPs", "algorithmic_equiv_TFLOPs", "dominant_kernel",
                                                                "achieved_vs_f32_roof", "note") if k in rr}
                del swl, sout
            except Exception as ex:  # noqa: BLE001
                entry["error"] = repr(ex)
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
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
