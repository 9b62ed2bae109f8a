"""Pins the numpy oracle against golden vectors produced by the REAL reference
(oracle/gen_golden.py, run in the build container against /root/reference)."""
import numpy as np
import pytest

from oracle import irsde_oracle as O


def relerr(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


def _sched_from_cfg(g, tag):
    ms, T, eps = g[tag + "/cfg"]
    return O.irsde_schedule(float(ms) if ms < 1 else int(ms), int(T), str(g[tag + "/sched"]), float(eps))


@pytest.mark.parametrize("tag", ["s10_T100", "s50_T100", "s50_T200", "s25_T100_lin", "s0p1_T50_const"])
def test_schedule_tables(golden, tag):
    g = golden.schedule
    sch = _sched_from_cfg(g, tag)
    # fp32 tables: numpy's cosf/cumsum differ from torch's by 1 ulp and `1 - cos^2`, `1 - exp(-x)`
    # amplify that by cancellation at small t, hence the per-table tolerances.  (The product builds
    # its tables with the same torch CPU ops as the reference and is checked bit-exactly elsewhere.)
    assert abs(float(sch["dt"]) - float(g[tag + "/dt"])) <= 1e-6 * abs(float(g[tag + "/dt"]))
    for n, rtol in (("thetas", 3e-5), ("sigmas", 2e-5), ("thetas_cumsum", 1e-5), ("sigma_bars", 5e-4)):
        ref = g[tag + "/" + n]
        # index 0 of the cosine theta table is ~1e-8 cancellation noise and never used (sde_utils.py:81-83)
        np.testing.assert_allclose(sch[n][1:], ref[1:], rtol=rtol, atol=1e-9, err_msg=n)
    T = sch["T"]
    for t in range(1, T + 1):
        g0, t1, t2, std = O.posterior_coeffs(sch, t)
        np.testing.assert_allclose(g0, g[tag + "/x0_gain"][t], rtol=2e-5)
        # (1-C^2)/(1-B^2) is a ratio of two cancellations in fp32: ill-conditioned at small t
        np.testing.assert_allclose(t1, g[tag + "/post_term1"][t], rtol=2e-3, atol=1e-6)
        np.testing.assert_allclose(t2, g[tag + "/post_term2"][t], rtol=2e-3, atol=1e-6)
        np.testing.assert_allclose(std, g[tag + "/post_std"][t], rtol=2e-3, atol=1e-12)
    # away from the ill-conditioned first steps the coefficients agree tightly
    for t in range(8, T + 1):
        g0, t1, t2, std = O.posterior_coeffs(sch, t)
        np.testing.assert_allclose(t1, g[tag + "/post_term1"][t], rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(t2, g[tag + "/post_term2"][t], rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(std, g[tag + "/post_std"][t], rtol=2e-4, atol=1e-12)


def test_schedule_anchors_baseline_md():
    """BASELINE.md §4 known-answer anchors."""
    s = O.irsde_schedule(10, 100, "cosine", 0.005)
    np.testing.assert_allclose(s["dt"], 0.104093805, rtol=1e-6)
    np.testing.assert_allclose(s["thetas"][1], 0.00169456005, rtol=1e-5)
    np.testing.assert_allclose(s["thetas"][100], 0.999766588, rtol=1e-6)
    np.testing.assert_allclose(s["sigmas"][100], 0.0554528832, rtol=1e-6)
    np.testing.assert_allclose(s["thetas_cumsum"][100], 50.8994484, rtol=1e-6)
    np.testing.assert_allclose(s["sigma_bars"][1], 0.000736524758, rtol=1e-5)
    np.testing.assert_allclose(s["sigma_bars"][100], 0.0392151959, rtol=1e-6)
    np.testing.assert_allclose(O.posterior_coeffs(s, 2)[3], 5.96858503e-4, rtol=1e-4)
    np.testing.assert_allclose(O.posterior_coeffs(s, 100)[3], 1.69992093e-2, rtol=1e-5)
    s = O.irsde_schedule(50, 200, "cosine", 0.005)
    np.testing.assert_allclose(s["dt"], 0.052307323, rtol=1e-6)
    np.testing.assert_allclose(s["thetas_cumsum"][200] * s["dt"], 5.298317, rtol=1e-6)


@pytest.mark.parametrize("tag", ["s10_T100", "s50_T200"])
def test_single_steps(golden, tag):
    g = golden.steps
    ms, T = (10, 100) if tag == "s10_T100" else (50, 200)
    sch = O.irsde_schedule(ms, T, "cosine", 0.005)
    gs = golden.schedule
    sch.update({n: gs[tag + "/" + n] for n in ("thetas", "sigmas", "thetas_cumsum", "sigma_bars",
                                               "post_term1", "post_term2", "post_std", "x0_gain")},
               dt=gs[tag + "/dt"])
    x, mu, eh = g[tag + "/x"], g[tag + "/mu"], g[tag + "/eps_hat"]
    z = O.synth_noise(5, T, x.shape)
    for t in (1, 2, T // 2, T):
        for dtype, tol in ((np.float32, 2e-5), (np.float64, 2e-5)):
            np.testing.assert_allclose(O.reverse_sde_step(sch, x, mu, eh, z[t], t, dtype),
                                       g[tag + "/sde_t%d" % t], rtol=tol, atol=tol)
            np.testing.assert_allclose(O.reverse_ode_step(sch, x, mu, eh, t, dtype),
                                       g[tag + "/ode_t%d" % t], rtol=tol, atol=tol)
            np.testing.assert_allclose(O.reverse_posterior_step(sch, x, mu, eh, z[t], t, dtype),
                                       g[tag + "/post_t%d" % t], rtol=tol, atol=tol)


def test_param_inventory_matches_reference_count():
    assert len(O.unet_param_shapes(3, 3, 64, 4)) == 151
    n = sum(int(np.prod(s)) for s in O.unet_param_shapes(3, 3, 64, 4).values())
    assert n == 137147523


@pytest.mark.parametrize("tag", ["nf32d2_2x24x20", "nf64d4_1x64x64", "nf64d4_2x40x56"])
def test_unet_forward(golden, tag):
    g = golden.forward
    nf, depth, B, H, W = (int(v) for v in g[tag + "/cfg"])
    params = O.synth_params(seed=0, nf=nf, depth=depth)
    lq, xT = O.synth_inputs(1234, B, H, W)
    for t in g[tag + "/ts"]:
        ref = g[tag + "/t%d" % t]
        y = O.unet_forward(params, xT, lq, int(t), depth=depth, dtype=np.float64)
        err = np.abs(y - ref).max() / np.abs(ref).max()
        assert err < 2e-5, (tag, t, err)
    if tag == "nf32d2_2x24x20":
        y = O.unet_forward(params, xT, lq, np.array([5, 60]), depth=depth, dtype=np.float64)
        ref = g[tag + "/tvec"]
        assert np.abs(y - ref).max() / np.abs(ref).max() < 2e-5
        # fp32 oracle agrees too (looser)
        y32 = O.unet_forward(params, xT, lq, 3, depth=depth, dtype=np.float32)
        ref = g[tag + "/t3"]
        assert np.abs(y32 - ref).max() / np.abs(ref).max() < 1e-4


@pytest.mark.parametrize("mode", ["sde", "ode", "posterior"])
def test_sampler_small(golden, mode):
    g = golden.sampler
    tag = "nf32d2_2x16x16_T20"
    nf, depth, B, H, W, T = (int(v) for v in g[tag + "/cfg"])
    params = O.synth_params(seed=0, nf=nf, depth=depth)
    lq, xT = O.synth_inputs(1234, B, H, W)
    z = O.synth_noise(7, T, (B, 3, H, W))
    sch = O.irsde_schedule(10, T, "cosine", 0.005)
    y = O.sample(params, sch, xT, lq, mode, noise=z, depth=depth, dtype=np.float64)
    ref = g[tag + "/" + mode]
    err = np.abs(y - ref).max() / np.abs(ref).max()
    # the reverse drift expands perturbations ~200x: fp32 reference vs fp64 oracle
    assert err < 1e-3, (mode, err)


def test_philox_known_answer():
    """Random123 known-answer vectors for philox4x32-10 (kat_vectors: zero / pi inputs)."""
    out = O.philox4x32_10(np.zeros((1, 4), dtype=np.uint32), (0, 0))[0]
    assert [hex(v) for v in out] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    ctr = np.array([[0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344]], dtype=np.uint32)
    out = O.philox4x32_10(ctr, (0xa4093822, 0x299f31d0))[0]
    assert [hex(v) for v in out] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_device_normal_moments():
    z = O.device_normal(1234, 7, 3, 3 * 64 * 64)
    assert abs(z.mean()) < 0.03 and abs(z.std() - 1) < 0.03


@pytest.mark.parametrize("tag", ["nf32d2_2x24x20", "nf64d4_2x40x56"])
def test_torch_cpu_port_matches_reference(golden, tag):
    """oracle/torch_cpu_port.py (the cpu_baseline 'port') reproduces the real reference's outputs."""
    import torch
    from oracle import torch_cpu_port as TP
    g = golden.forward
    nf, depth, B, H, W = (int(v) for v in g[tag + "/cfg"])
    params = {k: torch.from_numpy(v) for k, v in O.synth_params(seed=0, nf=nf, depth=depth).items()}
    lq, xT = O.synth_inputs(1234, B, H, W)
    for t in g[tag + "/ts"]:
        with torch.no_grad():
            y = TP.unet_forward(params, torch.from_numpy(xT), torch.from_numpy(lq), int(t), depth).numpy()
        ref = g[tag + "/t%d" % t]
        assert np.abs(y - ref).max() / np.abs(ref).max() < 1e-5


def test_torch_cpu_port_matches_reference_at_256(golden):
    """The cpu_baseline port at the benchmarked image size (1x3x256x256) and at the Rain100H size (reflect pad 321 -> 336,
    481 -> 496) vs the REAL reference (tests/golden/fullres.npz; `sub3` = every third pixel, oracle/gen_golden.py)."""
    import torch
    from oracle import torch_cpu_port as TP
    g = golden.fullres
    params = {k: torch.from_numpy(v) for k, v in O.synth_params(seed=0, nf=64, depth=4).items()}
    lq, xT = O.synth_inputs(1234, 1, 256, 256)
    with torch.no_grad():
        y = TP.unet_forward(params, torch.from_numpy(xT), torch.from_numpy(lq), 50, 4).numpy()
    ref = g["unet_1x256x256/t50"]
    assert np.abs(y - ref).max() / np.abs(ref).max() < 1e-5
    lq, xT = O.synth_inputs(1234, 2, 321, 481)
    with torch.no_grad():
        y = TP.unet_forward(params, torch.from_numpy(xT[:1]), torch.from_numpy(lq[:1]), 37, 4).numpy()
    ref = g["unet_2x321x481/t37_bottom"][:1]
    assert np.abs(y[:, :, -20:, :] - ref).max() / np.abs(ref).max() < 1e-5
    ref = g["unet_2x321x481/t37_sub3"][:1]
    assert np.abs(y[..., 1::3, 2::3] - ref).max() / np.abs(ref).max() < 1e-5


def test_f16_restatement_rounds_like_ieee_binary16():
    """O.round_f16 / O.f16_convs (the checker of IRSDE_FLAG_FP16): RNE to 11 significand bits, ties to even, and the
    conv oracle applies it to both operands."""
    a = np.array([1.0, 1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11, 1.0 + 2.0 ** -10, 65504.0, 1e-8, -0.1], dtype=np.float32)
    r = O.round_f16(a)
    assert r.dtype == np.float32
    assert r[0] == 1.0 and r[1] == 1.0 and r[2] == np.float32(1.0 + 2.0 ** -9) and r[3] == np.float32(1.0 + 2.0 ** -10)
    assert r[4] == 65504.0 and r[5] == np.float32(np.float16(1e-8)) and r[6] == np.float32(np.float16(-0.1))
    rs = np.random.RandomState(0)
    x = rs.standard_normal((1, 8, 6, 6)).astype(np.float32)
    w = rs.standard_normal((4, 8, 3, 3)).astype(np.float32)
    with O.f16_convs():
        y16 = O.conv2d(x, w, pad=1)
    assert np.allclose(y16, O.conv2d(O.round_f16(x), O.round_f16(w), pad=1), rtol=0, atol=0)
    assert 1e-5 < np.abs(y16 - O.conv2d(x, w, pad=1)).max() < 5e-2
    assert not O.CONV_OPERANDS_F16


def test_torch_cpu_port_sampler_steps(golden):
    import torch
    from oracle import torch_cpu_port as TP
    g = golden.sampler
    tag = "nf32d2_2x16x16_T20"
    nf, depth, B, H, W, T = (int(v) for v in g[tag + "/cfg"])
    params = {k: torch.from_numpy(v) for k, v in O.synth_params(seed=0, nf=nf, depth=depth).items()}
    lq, xT = O.synth_inputs(1234, B, H, W)
    z = torch.from_numpy(O.synth_noise(7, T, (B, 3, H, W)))
    sch = O.irsde_schedule(10, T, "cosine", 0.005)
    y = TP.reverse_sde_steps(params, sch, torch.from_numpy(xT), torch.from_numpy(lq), z, T, T, depth).numpy()
    ref = g[tag + "/sde"]
    assert np.abs(y - ref).max() / np.abs(ref).max() < 1e-3


NAF_CFGS = {"refusion": dict(width=64, enc_blk_nums=(1, 1, 1, 28), middle_blk_num=1, dec_blk_nums=(1, 1, 1, 1)),
            "w32_e12": dict(width=32, enc_blk_nums=(1, 2), middle_blk_num=1, dec_blk_nums=(1, 1))}


@pytest.mark.parametrize("tag", ["w32_e12_2x24x20", "refusion_2x40x56"])
def test_nafnet_forward(golden, tag):
    """ConditionalNAFNet (Refusion) oracle vs the real reference (zero-pad path included)."""
    g = golden.nafnet
    cfg = NAF_CFGS["refusion" if tag.startswith("refusion") else "w32_e12"]
    B, H, W = (int(v) for v in g[tag + "/shape"])
    params = O.naf_synth_params(seed=0, img_channel=3, **cfg)
    assert len(O.naf_param_shapes(3, **NAF_CFGS["refusion"])) == 668
    lq, xT = O.synth_inputs(1234, B, H, W, max_sigma=50)
    for t in g[tag + "/ts"]:
        y = O.nafnet_forward(params, xT, lq, int(t), cfg["enc_blk_nums"], cfg["middle_blk_num"], cfg["dec_blk_nums"])
        ref = g[tag + "/t%d" % t]
        assert np.abs(y - ref).max() / np.abs(ref).max() < 2e-5, (tag, t)


def test_dsde_schedule_and_unet(golden):
    """denoising-sde variant: DenoisingSDE tables / optimal timestep and the unconditional UNet with full attention."""
    g = golden.dsde
    sch = O.dsde_schedule(75, 100)
    assert abs(float(sch["dt"]) - float(g["sde/dt"])) <= 1e-6 * float(g["sde/dt"])
    for n, rtol in (("thetas", 3e-5), ("sigmas", 2e-5), ("thetas_cumsum", 1e-5), ("sigma_bars", 5e-4)):
        np.testing.assert_allclose(sch[n][1:], g["sde/" + n][1:], rtol=rtol, atol=1e-9, err_msg=n)
    ref_sch = dict(sch, **{n: g["sde/" + n] for n in ("thetas", "sigmas", "thetas_cumsum", "sigma_bars")}, dt=g["sde/dt"])
    assert [O.dsde_optimal_timestep(ref_sch, s) for s in (15, 25, 50)] == list(g["sde/opt_t"])
    for tag in ("nf32d2_2x24x20", "nf64d4_2x88x80"):
        nf, depth, B, H, W = (int(v) for v in g[tag + "/cfg"])
        params = O.uncond_synth_params(seed=0, nf=nf, depth=depth)
        _, xT = O.synth_inputs(1234, B, H, W, max_sigma=25)
        for t in g[tag + "/ts"]:
            y = O.uncond_unet_forward(params, xT, int(t), depth=depth)
            ref = g[tag + "/t%d" % t]
            assert np.abs(y - ref).max() / np.abs(ref).max() < 2e-5, (tag, t)


def test_dsde_sampler_small(golden):
    g = golden.dsde
    tag, key = "nf32d2_2x24x20", "nf32d2_2x24x20/sampler_2x16x16"
    params = O.uncond_synth_params(seed=0, nf=32, depth=2)
    T = int(g[key + "/T"])
    gs = golden.dsde
    sch = O.dsde_schedule(75, 100)
    sch.update({n: gs["sde/" + n] for n in ("thetas", "sigmas", "thetas_cumsum", "sigma_bars")}, dt=gs["sde/dt"])
    z = O.synth_noise(7, 100, (2, 3, 16, 16))
    for mode in ("ode", "sde"):
        y = O.dsde_sample(params, sch, g[key + "/noisy"], mode == "ode", noise=z, depth=2, T=T)
        ref = g[key + "/" + mode]
        assert np.abs(y - ref).max() / np.abs(ref).max() < 1e-3, mode


@pytest.mark.parametrize("tag", ["rgb_1x3x40x52_cb0", "rgb_2x3x64x48_cb4", "gray_1x1x33x37_cb0"])
def test_eval_tail_vs_reference(golden, tag):
    """tensor2img / PSNR / SSIM / Y-channel restatement against the reference's own functions (cv2's two calls stubbed
    with their definitions, oracle/gen_golden.py gen_metrics)."""
    g = golden.metrics
    B, C, H, W, cb = (int(v) for v in g[tag + "/cfg"])
    out, gt, want = g[tag + "/out"], g[tag + "/gt"], g[tag + "/metrics"]
    assert np.array_equal(O.tensor2img(out[0]), g[tag + "/img0"])
    for b in range(B):
        got = np.array(O.eval_tail(out[b], gt[b], cb))
        k = 4 if C == 3 else 2
        np.testing.assert_allclose(got[:k], want[b, :k], rtol=1e-10, atol=0)
        assert C == 3 or np.isnan(got[2:]).all()


@pytest.mark.parametrize("tag,cfg", [("nasde_1x3x40x52", dict(ch=8, ch_mult=(4, 8, 8, 16), embed_dim=8)),
                                     ("bokeh_2x3x24x32", dict(ch=16, ch_mult=(1, 2, 4), embed_dim=4))])
def test_latent_unet_vs_reference(golden, tag, cfg):
    """Latent UNet encode / decode restatement vs the reference module (latent-dehazing UNet_arch.py)."""
    g = golden.latent
    B, H, W = (int(v) for v in g[tag + "/shape"])
    depth = len(cfg["ch_mult"])
    params = O.latent_unet_synth_params(seed=0, in_ch=3, out_ch=3, **cfg)
    lq, _ = O.synth_inputs(1234, B, H, W)
    lat, hid = O.latent_unet_encode(params, lq, depth)
    assert relerr(lat, g[tag + "/latent"]) < 2e-5
    assert len(hid) == 2 * depth + 1
    for i, h in enumerate(hid):
        assert h.shape == g[tag + "/hidden%d" % i].shape and relerr(h, g[tag + "/hidden%d" % i]) < 2e-5, i
    assert relerr(O.latent_unet_decode(params, lat, hid, depth, H, W), g[tag + "/decode"]) < 2e-5
    hid_ref = [g[tag + "/hidden%d" % i] for i in range(len(hid))]
    assert relerr(O.latent_unet_decode(params, g[tag + "/latent2"], hid_ref, depth, H, W), g[tag + "/decode2"]) < 2e-5


def test_latent_nafnet_and_pipeline_vs_reference(golden):
    """latent-task ConditionalNAFNet (ending(x + intro)) and the encode -> reverse_sde/ode in the latent -> decode pipeline."""
    g = golden.latent
    nparams = O.naf_synth_params(seed=0, img_channel=8, width=32, middle_blk_num=1, enc_blk_nums=(1, 2), dec_blk_nums=(1, 1))
    kw = dict(enc_blk_nums=(1, 2), middle_blk_num=1, dec_blk_nums=(1, 1), intro_skip=True)
    for t in (4, 61):
        y = O.nafnet_forward(nparams, g["naf/xt"], g["naf/cond"], t, **kw)
        assert relerr(y, g["naf/t%d" % t]) < 2e-5
    uparams = O.latent_unet_synth_params(seed=0, in_ch=3, out_ch=3, ch=8, ch_mult=(4, 8, 8, 16), embed_dim=8)
    lq, _ = O.synth_inputs(1234, 1, 40, 52)
    lat, hid = O.latent_unet_encode(uparams, lq, 4)
    T = int(g["pipe/T"])
    sch = O.irsde_schedule(50, T, "cosine", 0.005)
    noisy = lat + g["pipe/z0"] * sch["max_sigma"]
    z = O.synth_noise(7, T, lat.shape)
    net = lambda x, mu, t: O.nafnet_forward(nparams, x, mu, t, **kw)  # noqa: E731
    for mode in ("sde", "ode"):
        x0 = O.sample(nparams, sch, noisy, lat, mode, noise=z, net=net)
        assert relerr(x0, g["pipe/latent_" + mode]) < 2e-3
        assert relerr(O.latent_unet_decode(uparams, x0, hid, 4, 40, 52), g["pipe/out_" + mode]) < 2e-3


def test_latent_bokeh_nafnet_vs_reference(golden):
    """Lens-conditioned ConditionalNAFNet (latent-bokeh): forward with per-image lens triples and the kwargs sampler path."""
    g = golden.latent
    bp = O.naf_synth_params(seed=3, img_channel=4, width=32, middle_blk_num=1, enc_blk_nums=(1, 2), dec_blk_nums=(1, 1), lens=True)
    kw = dict(enc_blk_nums=(1, 2), middle_blk_num=1, dec_blk_nums=(1, 1))
    lens = g["bokeh/lens"]
    y = O.nafnet_forward(bp, g["bokeh/xt"], g["bokeh/cond"], np.array([5, 60]), lens_info=[lens[:, i] for i in range(3)], **kw)
    assert relerr(y, g["bokeh/tvec"]) < 2e-5
    li1 = [lens[:1, i] for i in range(3)]
    y = O.nafnet_forward(bp, g["bokeh/xt"][:1], g["bokeh/cond"][:1], 33, lens_info=li1, **kw)
    assert relerr(y, g["bokeh/t33"]) < 2e-5
    T = int(g["bokeh/T"])
    sch = O.irsde_schedule(50, T, "cosine", 0.005)
    net = lambda x, mu, t: O.nafnet_forward(bp, x, mu, t, lens_info=li1, **kw)  # noqa: E731
    x0 = O.sample(bp, sch, g["bokeh/xt"][:1], g["bokeh/cond"][:1], "sde", noise=O.synth_noise(9, T, (1, 4, 12, 10)), net=net)
    assert relerr(x0, g["bokeh/sde"]) < 2e-3


def test_fused_attention_16bit_restatement():
    """O.attn_block_fused16 (the checker of the engine's fused attention kernels in the 16-bit operand modes, r05): stays within the operand-rounding noise of
    the unrounded block, fp16 closer than bf16, and — with to_out scaled so that the attention core dominates the block — fixes the fp16 tensor plan's
    loss of the v / N context in fp16's subnormal range (scaled by 2^ceil(log2 N) through the 16-bit products)."""
    params = O.synth_params(seed=0, nf=64, depth=4)
    p = {k: np.asarray(v, dtype=np.float64) for k, v in params.items()}
    pref, n = "downs.0.2.", 64 * 64
    p[pref + "fn.fn.to_out.0.weight"] = p[pref + "fn.fn.to_out.0.weight"] * n
    x = np.random.RandomState(0).standard_normal((1, 64, 64, 64))
    full = O.attn_block(p, pref, x) - x
    scale = np.abs(full).max()
    assert scale > 0.5
    e_bf = np.abs(O.attn_block_fused16(p, pref, x) - x - full).max() / scale
    e_f16 = np.abs(O.attn_block_fused16(p, pref, x, f16=True) - x - full).max() / scale
    with O.f16_convs():
        e_tensor_f16 = np.abs(O.attn_block(p, pref, x) - x - full).max() / scale
        with O.fused_attn_16():
            assert np.array_equal(O.attn_block(p, pref, x), O.attn_block_fused16(p, pref, x, f16=True))
    assert 1e-5 < e_bf < 2e-3 and e_f16 < e_bf / 4 and e_f16 < e_tensor_f16 / 10, (e_bf, e_f16, e_tensor_f16)
    st = O.attn_block_fused16(p, pref, O.round_bf16(x), store_bf16=True)
    assert np.array_equal(st, O.round_bf16(st))


@pytest.mark.parametrize("tag", ["nf64d4_1x32x32", "nf64d4_1x64x64", "nf64d4_2x40x56"])
def test_unet_forward_attention_sensitive(golden, tag):
    """r05: the REAL reference with attention-sensitive weights (O.attn_sensitive_params; oracle/gen_golden.py::gen_forward_attn).  With the default synthetic
    weights to_out's bias dominates every LinearAttention block and the other forward goldens would not notice a wrong softmax / context / output product; here
    the network output moves by 0.4 - 0.46 of its maximum when the attention branches change, and the oracle follows the reference to 1e-6 — whole network and, for
    the small case, block by block on the reference's own block inputs (forward hooks)."""
    g = golden.forward_attn
    nf, depth, B, H, W, t = (int(v) for v in g[tag + "/cfg"])
    base = O.synth_params(seed=0, nf=nf, depth=depth)
    params = O.attn_sensitive_params(base, H, W, depth)
    lq, xT = O.synth_inputs(1234, B, H, W)
    ref = g[tag + "/y"]
    y = O.unet_forward(params, xT, lq, t, depth=depth, dtype=np.float64)
    assert np.abs(y - ref).max() / np.abs(ref).max() < 2e-5
    if tag == "nf64d4_1x32x32":
        y0 = O.unet_forward(base, xT, lq, t, depth=depth, dtype=np.float64)
        assert np.abs(y0 - ref).max() / np.abs(ref).max() > 0.1          # the fixture is attention-sensitive
        p = {k: np.asarray(v, dtype=np.float64) for k, v in params.items()}
        for name in ("downs.0.2", "downs.2.2", "mid_attn"):
            x, out = g[tag + "/" + name + "/in"].astype(np.float64), g[tag + "/" + name + "/out"].astype(np.float64)
            branch = np.abs(out - x).max()
            assert branch > 0.5 and np.abs(O.attn_block(p, name + ".", x) - out).max() / branch < 1e-5, name
