"""Host-side logic of the product package (no GPU): schedule/coefficient tables bit-identical to the
reference's, reference-compatible state_dict, loud failure without a GPU, sharding arithmetic."""
import numpy as np
import pytest
import torch

import image_restoration_sde_amd as P
from oracle import irsde_oracle as O

CFGS = {"s10_T100": (10, 100, "cosine", 0.005), "s50_T100": (50, 100, "cosine", 0.005),
        "s50_T200": (50, 200, "cosine", 0.005), "s25_T100_lin": (25, 100, "linear", 0.005),
        "s0p1_T50_const": (0.1, 50, "constant", 0.01)}


@pytest.mark.parametrize("tag", list(CFGS))
def test_schedule_bit_identical_to_reference(golden, tag):
    ms, T, sched, eps = CFGS[tag]
    g = golden.schedule
    sde = P.IRSDE(ms, T, sched, eps, device="cpu")
    assert float(sde.dt) == float(g[tag + "/dt"])
    for n in ("thetas", "sigmas", "thetas_cumsum", "sigma_bars"):
        assert np.array_equal(getattr(sde, n).numpy(), g[tag + "/" + n]), n
    c = sde._coef.numpy()
    assert c.shape == (T + 1, 12)
    assert np.array_equal(c[1:, 5], g[tag + "/x0_gain"][1:])
    assert np.array_equal(c[1:, 6], g[tag + "/post_term1"][1:])
    assert np.array_equal(c[1:, 7], g[tag + "/post_term2"][1:])
    assert np.array_equal(c[1:, 8], g[tag + "/post_std"][1:])
    assert np.array_equal(c[:, 0], g[tag + "/thetas"]) and np.array_equal(c[:, 2], g[tag + "/sigma_bars"])


def test_denoising_sde_tables_bit_identical(golden):
    g = golden.dsde
    sde = P.DenoisingSDE(max_sigma=75, T=100, device="cpu")
    assert float(sde.dt) == float(g["sde/dt"])
    for n in ("thetas", "sigmas", "thetas_cumsum", "sigma_bars"):
        assert np.array_equal(getattr(sde, n).numpy(), g["sde/" + n]), n
    assert [int(sde.get_optimal_timestep(s)) for s in (15, 25, 50)] == [int(v) for v in g["sde/opt_t"]]
    c = sde._coef.numpy()
    assert np.array_equal(c[:, 9], np.exp(np.float32(-2) * g["sde/thetas_cumsum"] * np.float32(g["sde/dt"])).astype(np.float32)) or \
        np.allclose(c[:, 9], np.exp(-2.0 * g["sde/thetas_cumsum"].astype(np.float64) * float(g["sde/dt"])), rtol=1e-6)
    assert P.DenoisingSDE(1, 10).max_sigma == 1 and abs(P.DenoisingSDE(25, 10).max_sigma - 25 / 255) < 1e-12  # '>' not '>='


def test_variant_state_dicts_are_reference_compatible():
    sd = P.denoising_sde.ConditionalUNet(3, 3, 64, 4).state_dict()
    sh = O.uncond_unet_param_shapes(3, 3, 64, 4)
    assert set(sd) == set(sh) and all(tuple(sd[k].shape) == tuple(sh[k]) for k in sh)
    sd = P.ConditionalNAFNet(3, width=64, enc_blk_nums=[1, 1, 1, 28], middle_blk_num=1, dec_blk_nums=[1, 1, 1, 1]).state_dict()
    sh = O.naf_param_shapes(3, 64, 1, (1, 1, 1, 28), (1, 1, 1, 1))
    assert set(sd) == set(sh) and all(tuple(sd[k].shape) == tuple(sh[k]) for k in sh)


def test_irsde_surface_matches_reference_names():
    sde = P.IRSDE(10, 100, "cosine", 0.005, device="cpu")
    for name in ["set_mu", "set_model", "noise_state", "reverse_sde", "reverse_ode", "reverse_posterior",
                 "generate_random_states", "noise_fn", "score_fn", "get_score_from_noise", "reverse_sde_step_mean",
                 "reverse_optimum_step", "reverse_optimum_std", "weights", "mu_bar", "sigma_bar", "forward",
                 "drift", "dispersion", "sde_reverse_drift", "ode_reverse_drift", "get_init_state_from_noise",
                 "reverse_posterior_step", "get_real_noise", "get_real_score", "optimal_reverse"]:
        assert callable(getattr(sde, name)), name
    for attr in ["T", "dt", "max_sigma", "thetas", "sigmas", "thetas_cumsum", "sigma_bars", "mu", "model"]:
        assert hasattr(sde, attr)
    assert abs(sde.max_sigma - 10 / 255) < 1e-12
    assert P.IRSDE(0.5, 10).max_sigma == 0.5  # sde_utils.py:86
    with pytest.raises(ValueError):
        P.IRSDE(10, 10, schedule="quadratic")


def test_training_helpers_match_oracle_formulas(golden):
    """The helper formulas reproduce the reference's single-step goldens (elementwise torch algebra)."""
    g = golden.steps
    sde = P.IRSDE(10, 100, "cosine", 0.005, device="cpu")
    x, mu, eh = (torch.from_numpy(g["s10_T100/" + k]) for k in ("x", "mu", "eps_hat"))
    sde.set_mu(mu)
    for t in (1, 2, 50, 100):
        score = sde.get_score_from_noise(eh, t)
        np.testing.assert_allclose(sde.reverse_ode_step(x, score, t).numpy(), g["s10_T100/ode_t%d" % t], rtol=1e-6,
                                   atol=1e-6)
        x0 = sde.get_init_state_from_noise(x, eh, t)
        mean = sde.reverse_optimum_step(x, x0, t)
        z = torch.from_numpy(O.synth_noise(5, 100, x.shape)[t])
        np.testing.assert_allclose((mean + sde.reverse_optimum_std(t) * z).numpy(), g["s10_T100/post_t%d" % t],
                                   rtol=1e-6, atol=1e-6)


def test_unet_state_dict_is_reference_compatible():
    m = P.ConditionalUNet(in_nc=3, out_nc=3, nf=64, depth=4)
    sd = m.state_dict()
    shapes = O.unet_param_shapes(3, 3, 64, 4)
    assert set(sd) == set(shapes) and len(sd) == 151
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    assert sum(v.numel() for v in sd.values()) == 137147523
    # loads a reference-style checkpoint (with DataParallel "module." prefixes) through DenoisingModel's loader
    params = O.synth_params(seed=0, nf=32, depth=2)
    m2 = P.ConditionalUNet(3, 3, 32, depth=2)
    m2.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    assert np.array_equal(m2.state_dict()["ups.0.3.1.bias"].numpy(), params["ups.0.3.1.bias"])


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU instead of computing on the CPU."""
    m = P.ConditionalUNet(3, 3, 32, depth=2)
    x = torch.zeros(1, 3, 16, 16)
    with pytest.raises(P.IrsdeError):
        m(x, x, 5)
    sde = P.IRSDE(10, 10, "cosine", 0.005, device="cpu")
    sde.set_model(m)
    sde.set_mu(x)
    for fn in (sde.reverse_sde, sde.reverse_ode, sde.reverse_posterior):
        with pytest.raises(P.IrsdeError):
            fn(x)


def test_missing_library_is_loud(monkeypatch, tmp_path):
    from image_restoration_sde_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.IrsdeLibraryError):
        _lib.lib()


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under the product package may import or execute it."""
    import os
    import re
    root = os.path.dirname(P.__file__)
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|oracle[./]irsde_oracle|#include\s+\".*oracle", re.M)
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                assert not pat.search(open(os.path.join(dp, f)).read()), f


def test_shard_bounds():
    for n in (1, 7, 8, 16, 17):
        for w in (1, 2, 4, 8):
            cover = []
            for r in range(w):
                lo, hi = P.shard_bounds(n, w, r)
                assert 0 <= lo <= hi <= n
                cover += list(range(lo, hi))
            assert cover == list(range(n))
    assert P.shard_bounds(16, 8, 3) == (6, 8)
    with pytest.raises(ValueError):
        P.shard_bounds(4, 2, 2)


def test_create_model_boundary_without_gpu():
    opt = {"model": "denoising", "is_train": False, "gpu_ids": [0], "dist": False,
           "network_G": {"which_model_G": "ConditionalUNet", "setting": {"in_nc": 3, "out_nc": 3, "nf": 32, "depth": 2}},
           "path": {"pretrain_model_G": None, "strict_load": True}}
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    with pytest.raises(Exception):
        P.create_model(opt)  # .to('cuda') fails loudly on a CPU-only box; no silent CPU model
    assert P.define_G(opt).__class__.__name__ == "ConditionalUNet"


def test_latent_modules_are_reference_compatible():
    """latent.UNet / latent.ConditionalNAFNet own exactly the reference's parameters (nasde.yml: 69 and 668 tensors)."""
    m = P.latent.UNet(in_ch=3, out_ch=3, ch=8, ch_mult=[4, 8, 8, 16], embed_dim=8)
    sd, sh = m.state_dict(), O.latent_unet_param_shapes(3, 3, 8, (4, 8, 8, 16), 8)
    assert len(sd) == 69 and set(sd) == set(sh) and all(tuple(sd[k].shape) == tuple(sh[k]) for k in sh)
    m = P.latent.UNet(in_ch=3, out_ch=3, ch=64, ch_mult=[1, 2, 4], embed_dim=4)   # latent-bokeh refusion.yml network_L
    sh = O.latent_unet_param_shapes(3, 3, 64, (1, 2, 4), 4)
    assert set(m.state_dict()) == set(sh)
    n = P.latent.ConditionalNAFNet(img_channel=8, width=64, enc_blk_nums=[1, 1, 1, 28], middle_blk_num=1, dec_blk_nums=[1, 1, 1, 1])
    sh = O.naf_param_shapes(8, 64, 1, (1, 1, 1, 28), (1, 1, 1, 1))
    sd = n.state_dict()
    assert len(sd) == 668 and set(sd) == set(sh) and all(tuple(sd[k].shape) == tuple(sh[k]) for k in sh)
    with pytest.raises(P.IrsdeError):
        P.latent.UNet().decode(torch.zeros(1, 4, 8, 8), [])       # decode before encode / CPU tensors: loud, no fallback
    with pytest.raises(P.IrsdeError):
        P.latent.UNet().encode(torch.zeros(1, 3, 16, 16))


def test_compute_dtype_flags_and_metrics_fail_loudly_on_cpu():
    from image_restoration_sde_amd import _lib
    m = P.ConditionalUNet(3, 3, 32, depth=2)
    assert m.engine_flags == 0
    m.set_compute_dtype("bf16")
    assert m.engine_flags == _lib.FLAG_BF16
    m.set_compute_dtype("bf16_act")
    assert m.engine_flags == _lib.FLAG_BF16 | _lib.FLAG_BF16_ACT
    m.set_compute_dtype("fp32_split")   # r03: f32 everywhere, deep Winograd GEMMs on bf16 hi + lo pairs
    assert m.engine_flags == _lib.FLAG_SPLIT_BF16X2 == 16384
    m.set_compute_dtype("fp32_split_f16")   # the same path with fp16 pieces: fp32-equivalent per layer
    assert m.engine_flags == _lib.FLAG_SPLIT_F16X2 == 32768
    m.set_compute_dtype("fp16")
    assert m.engine_flags == _lib.FLAG_FP16
    m.set_compute_dtype("fp32")
    assert m.engine_flags == 0
    with pytest.raises(P.IrsdeError):
        m.set_compute_dtype("fp8")
    with pytest.raises(P.IrsdeError):
        P.metrics.evaluate_batch(torch.zeros(1, 3, 16, 16), torch.zeros(1, 3, 16, 16))
    with pytest.raises(P.IrsdeError):
        P.metrics.tensor2img(torch.zeros(3, 16, 16))


def _load_tool(name):
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", name + ".py")
    spec = importlib.util.spec_from_file_location("tool_" + name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_eval_folder_host_helpers(tmp_path):
    """tools/eval_folder.py: folder walk / LQ-GT pairing / image I/O / equal-size batching (the parts of the reference's
    test loop, deraining/test.py:93-217, that run on the host)."""
    import os
    from PIL import Image
    ef = _load_tool("eval_folder")
    rs = np.random.RandomState(0)
    lq, gt = tmp_path / "LQ", tmp_path / "GT" / "sub"
    lq.mkdir()
    gt.mkdir(parents=True)
    imgs = {}
    for i, (h, w) in enumerate([(20, 24), (20, 24), (16, 24), (20, 24)]):
        a = rs.randint(0, 256, size=(h, w, 3), dtype=np.uint8)
        imgs["im%d" % i] = a
        Image.fromarray(a).save(str(lq / ("im%d.png" % i)))
        Image.fromarray(255 - a).save(str(gt / ("im%d.PNG" % i)))
    (lq / "notes.txt").write_text("not an image")
    pairs = ef.pair_paths(str(lq), str(gt))
    assert [os.path.basename(p[0]) for p in pairs] == ["im0.png", "im1.png", "im2.png", "im3.png"]
    assert all(os.path.basename(p[1]).lower() == os.path.basename(p[0]) for p in pairs)
    x = ef.read_img(pairs[2][0])
    assert x.dtype == np.float32 and x.shape == (3, 16, 24)
    assert np.array_equal(x, imgs["im2"].transpose(2, 0, 1).astype(np.float32) / np.float32(255))   # RGB, CHW, [0, 1]
    sizes = [ef.read_img(p[0]).shape for p in pairs]
    assert ef.batches_of_equal_size(pairs, sizes, 16) == [[0, 1], [2], [3]]
    assert ef.batches_of_equal_size(pairs[:2], sizes[:2], 1) == [[0], [1]]
    # save_img takes the BGR uint8 image tensor2img produces and writes it as RGB
    ef.save_img(imgs["im0"][..., ::-1], str(tmp_path / "out" / "o.png"))
    assert np.array_equal(np.asarray(Image.open(str(tmp_path / "out" / "o.png"))), imgs["im0"])
    (gt / "extra.png").write_bytes((gt / "im0.PNG").read_bytes())
    with pytest.raises(ValueError):
        ef.pair_paths(str(lq), str(gt))
    with pytest.raises(FileNotFoundError):
        ef.list_images(str(tmp_path / "no_such_dir"))


def test_tuning_env_knobs_are_inert_without_IRSDE_TUNING():
    """A stray IRSDE_* tuning variable must not change the validated launch plan: the library reads them only through
    tuning_env_int(), which returns the default unless IRSDE_TUNING=1 (csrc/common.h)."""
    import os
    import re
    root = os.path.join(os.path.dirname(P.__file__), "csrc")
    for f in sorted(os.listdir(root)):
        if f.endswith((".hip", ".h")):
            src = open(os.path.join(root, f)).read()
            for m in re.finditer(r"getenv\(\"(\w+)\"\)", src):
                assert f == "common.h" and m.group(1) == "IRSDE_TUNING", (f, m.group(0))


def test_every_tuning_knob_is_documented():
    """DESIGN.md's "Tuning knobs" section lists exactly the knobs the code reads (tuning_env_int("IRSDE_...", default))."""
    import os
    import re
    root = os.path.join(os.path.dirname(P.__file__), "csrc")
    knobs = set()
    for f in sorted(os.listdir(root)):
        if f.endswith((".hip", ".h")):
            src = open(os.path.join(root, f)).read()
            knobs |= set(re.findall(r"tuning_env_int\(\"(IRSDE_\w+)\"", src))
            for m in re.finditer(r"tuning_env_int\(([^;]*?\?[^;]*?),", src):   # a name chosen by a conditional expression
                knobs |= set(re.findall(r"\"(IRSDE_\w+)\"", m.group(1)))
    design = open(os.path.join(os.path.dirname(os.path.dirname(P.__file__)), "DESIGN.md")).read()
    sec = design[design.index("### Tuning knobs"):design.index("### Sampler loop")]
    # the section abbreviates families: `IRSDE_X_MAXCIN` / `_MAXCOUT` — expand "/ `_SUFFIX`" against the preceding full name
    named = set(re.findall(r"`(IRSDE_\w+)`", sec))
    for full, rest in re.findall(r"`(IRSDE_\w+)`((?:\s*(?:\([^)]*\))?\s*/\s*`_\w+`)+)", sec):
        for suf in re.findall(r"`(_\w+)`", rest):
            named.add(full[:full.rindex("_")] + suf)
    assert knobs <= named, sorted(knobs - named)
    assert named - knobs <= {"IRSDE_TUNING"}, sorted(named - knobs)


def test_bench_roofline_object_from_an_op_profile():
    """bench.py's host logic (no GPU): the per-class parse of irsde_op_profile's text and the roofline object built from it —
    fractions are true fractions of the roof, the dominant class is the one with the largest time share, the split mode is
    labelled, the 16-bit modes quote the HBM roof first."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    text = ("   0.0284 ms  other\n"
            "   1.0000 ms  conv(winograd F4 fused) 16x64 T=65536 Cout=128 Cin=128 up=0 blocks=8192 flops=3.2e+11 exec=8.0e+10\n"
            "   0.5000 ms  conv(winograd F4 gemm x36) T=1024 Cout=1024 Cin=1024 flops=3.2e+11 exec=7.0e+10\n"
            "   0.2500 ms  conv(split bf16x2 winograd F4 gemm x36) T=1024 Cout=1024 Cin=1024 flops=3.2e+11 exec=7.0e+10\n"
            "   0.4000 ms  conv M=16384 Cout=1024 Cin=1536 k=1x1 s=1 up=0 splits=1 blocks=1024 flops=5.0e+10\n"
            "   0.0500 ms  wino_input T=1024 C=1024\n"
            "   0.3000 ms  linear_attention LayerNorm + k,v projection + context (fused) C=128\n"
            "   0.0200 ms  layernorm\n")
    cls = bench.op_classes(text)
    assert len(cls) == 8 and abs(sum(v[0] for v in cls.values()) - 2.5484) < 1e-9
    fused = [k for k in cls if k.startswith("wino4_fused64")][0]
    assert cls[fused] == [1.0, 8.0e10, 1]
    assert any(k.startswith("gemm_split2i_kernel") for k in cls)
    prof = {"conv_ms": 2.15, "wino_ms": 0.05, "conv_exec_flops": 2.7e11, "conv_flops": 1.01e12, "conv_launches": 4.0, "net_evals": 1.0,
            "conv_bytes": 1.0e9, "wall_ms": 2.55, "ln_ms": 0.02, "attn_ms": 0.3, "other_ms": 0.03}
    r = bench.roofline_object(prof, text, {"dtype": "fp32"})
    assert r["bound"] == "mfma" and r["peak"] == bench.PEAK_FP32_TFLOPS and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0 < r["frac"] <= r["mfma_kernel_frac"] <= 1
    assert r["dominant_kernel"]["name"] == fused and abs(r["dominant_kernel"]["executed_TFLOPs"] - 80.0) < 1e-6
    assert abs(sum(k["share"] for k in r["per_kernel"]) - 1.0) < 5e-3 and sum(k["n"] for k in r["per_kernel"]) == 8 == r["launches_per_evaluation"]
    assert abs(sum(k["share"] for k in r["per_kernel"] if "frac" in k) - (2.15 / 2.5484)) < 2e-3
    assert all(0 < k["frac"] <= 1 for k in r["per_kernel"] if "frac" in k and not k["name"].startswith("gemm_split"))
    # the whole line has to survive in the driver's 8 KB record: headline roofline + 7 secondaries of this size stay under 7 KB
    import json
    sec_entry = {"tag": bench.SECONDARY[0]["tag"], "value": 9.612, "unit": "images/s", "ms_per_step": 1664.2, "steps": 1, "ms_per_evaluation": 16.64,
                 "n_gpus": 1, "global_batch": 16, "mode": "reverse_ode", "T": 100, "dtype": "bf16_act",
                 "roofline": {k: r[k] for k in ("bound", "achieved", "unit", "frac", "mfma_kernel_frac", "hbm_frac", "whole_path_TFLOPs", "launches_per_evaluation")}}
    sec_entry["roofline"]["dominant_kernel"] = {k: r["dominant_kernel"][k] for k in ("name", "time_share", "frac")}
    line = {"metric": "restored images/sec at 256x256, 100-step IR-SDE reverse sampler", "value": 4.17, "roofline": r, "secondary": [sec_entry] * len(bench.SECONDARY),
            "config": {"workload": bench.workload_text(dict(model="unet", mode="sde", batch=16, size=256, T=100), 100)}}
    assert len(json.dumps(line, separators=(",", ":"))) < 6000   # + cpu_baseline (~600 B) + the scalar headline keys (~700 B) < 7 KB
    for dt in ("fp32_split", "fp32_split_f16"):   # two pipes: the fraction is taken against each kernel's own roof and stays <= 1
        rs_ = bench.roofline_object(prof, text, {"dtype": dt})
        assert "achieved_vs_f32_roof" in rs_ and rs_["peak"] is None and 0 < rs_["frac"] <= rs_["mfma_kernel_frac"] <= 1
        assert rs_["frac"] < r["frac"]   # the pair class is priced at the 16-bit roof, not at the f32 one
        # the op text describes ONE evaluation, the profile covers net_evals of them: the fraction does not depend on how many were timed
        prof3 = dict(prof, net_evals=3.0, **{k: 3 * prof[k] for k in ("conv_ms", "wino_ms", "conv_exec_flops", "conv_flops", "conv_launches", "conv_bytes",
                                                                      "wall_ms", "ln_ms", "attn_ms", "other_ms")})
        assert abs(bench.roofline_object(prof3, text, {"dtype": dt})["frac"] - rs_["frac"]) < 1e-9
    rb = bench.roofline_object(prof, text, {"dtype": "bf16_act"})
    assert rb["bound"] == "hbm" and rb["unit"] == "GB/s" and rb["peak"] == bench.PEAK_HBM_GBPS
    # every secondary workload names a model / dtype the Workload class knows, and the headline stays out of the list
    for w in bench.SECONDARY:
        assert w["model"] in ("unet", "nafnet", "latent", "dsde") and w["dtype"] in bench.DTYPE_LABEL and w["mode"] in ("sde", "ode", "posterior")
    assert not any((w["model"], w["dtype"], w["batch"], w["size"], w["T"], w["mode"]) == ("unet", "fp32", 16, 256, 100, "sde") for w in bench.SECONDARY)


def test_strong_scaling_shards_cover_the_global_batch():
    """`bench.py --scaling strong` / `sample_shard`: the shards of ONE global batch over 1 .. 8 ranks are contiguous, disjoint, cover
    the batch, differ by at most one image, and ranks beyond the batch get an empty shard (16 images over 8 GPUs = 2 each)."""
    for n in (1, 5, 16):
        for w in (1, 2, 3, 8):
            b = [P.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1 and sum(sizes) == n
    assert [P.shard_bounds(16, 8, r) for r in range(8)] == [(2 * r, 2 * r + 2) for r in range(8)]
