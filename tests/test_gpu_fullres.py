"""GPU parity at the shapes that are actually benchmarked (round 2): the production launch plan at 256x256 / 321x481 /
512x512 takes branches the small goldens never touch (Winograd F(4x4) on the 64-channel full-resolution level, the
tile-loop GEMM's components-per-block choice, the 24 576-block 1x1 path), so the REAL reference's outputs at those
shapes are committed (tests/golden/fullres.npz, oracle/gen_golden.py --only fullres) and compared here.

Also: batch independence of the production plan (B=16 vs 16 x B=1), the reduced-precision modes at 256x256, the N>1
launch path on real devices, the dataset evaluation driver, the T=0 / x0 corner semantics, and the seam with
reference-style foreign objects.  /root/reference is read only by the one test that is guarded on its presence.
"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import image_restoration_sde_amd as P
from image_restoration_sde_amd import _lib
from oracle import irsde_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def relerr(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


def sub3(y):
    return np.ascontiguousarray(y[..., 1::3, 2::3])   # oracle/gen_golden.py:sub3


_CACHE = {}


def unet64(prec="fp32"):
    key = "unet" if prec == "fp32" else "unet_" + prec
    if key not in _CACHE:
        params = O.synth_params(seed=0, nf=64, depth=4)
        m = P.ConditionalUNet(3, 3, 64, depth=4)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
        if prec != "fp32":
            m.set_compute_dtype(prec)
        _CACHE[key] = m.to(DEV).eval()
    return _CACHE[key]


def refusion_net(prec="fp32"):
    key = "naf" if prec == "fp32" else "naf_" + prec
    if key not in _CACHE:
        params = O.naf_synth_params(seed=0, img_channel=3, width=64, middle_blk_num=1, enc_blk_nums=(1, 1, 1, 28), dec_blk_nums=(1, 1, 1, 1))
        m = P.ConditionalNAFNet(img_channel=3, width=64, enc_blk_nums=[1, 1, 1, 28], middle_blk_num=1, dec_blk_nums=[1, 1, 1, 1])
        m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
        if prec != "fp32":
            m.set_compute_dtype(prec)
        _CACHE[key] = m.to(DEV).eval()
    return _CACHE[key]


# the plan-level checks below also run in the opt-in fp32_split_f16 mode (fp16 hi + lo operand pairs on the 16-bit MFMA pipe) inside the
# default `-m gpu` run, at the native tolerances
PRECS = ["fp32", "fp32_split_f16"]


# ---------------------------------------------------------------------------------------------
# reference goldens at benchmark shapes
# ---------------------------------------------------------------------------------------------
def test_unet_forward_256_vs_reference_golden(golden):
    """ConditionalUNet.forward at 1x3x256x256 (BASELINE configs[1] image size), t in {1, 50, 100}."""
    g = golden.fullres
    m = unet64()
    lq, xT = O.synth_inputs(1234, 1, 256, 256)
    x, c = torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV)
    for t in (1, 50, 100):
        y = m(x, c, t).cpu().numpy()
        ref = g["unet_1x256x256/t%d" % t]
        e = relerr(y if t == 50 else sub3(y), ref)
        print("unet 256x256 t=%d: %.3g" % (t, e))
        assert e < 1e-4, t


def test_unet_forward_321x481_vs_reference_golden(golden):
    """The Rain100H image size: 2x3x321x481, reflect pad 321 -> 336 and 481 -> 496."""
    g = golden.fullres
    m = unet64()
    lq, xT = O.synth_inputs(1234, 2, 321, 481)
    y = m(torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV), 37).cpu().numpy()
    assert y.shape == (2, 3, 321, 481)
    scale = np.abs(g["unet_2x321x481/t37_sub3"]).max()
    for got, key in ((sub3(y), "t37_sub3"), (y[:, :, -20:, :], "t37_bottom"), (y[:, :, :, -20:], "t37_right")):
        e = float(np.abs(got - g["unet_2x321x481/" + key]).max() / scale)
        print("unet 321x481 %s: %.3g" % (key, e))
        assert e < 1e-4, key


@pytest.mark.parametrize("mode", ["sde", "posterior"])
def test_sampler_256_vs_reference_golden(golden, mode):
    """Full T=100 reverse_sde / reverse_posterior at 1x3x256x256 with injected noise vs the REAL reference."""
    g = golden.fullres
    m = unet64()
    lq, xT = O.synth_inputs(1234, 1, 256, 256)
    z = O.synth_noise(7, 100, (1, 3, 256, 256))
    sde = P.IRSDE(10, 100, "cosine", 0.005, device=DEV)
    sde.set_model(m)
    sde.set_mu(torch.from_numpy(lq).to(DEV))
    sde.injected_noise = torch.from_numpy(z).to(DEV)
    fn = sde.reverse_sde if mode == "sde" else sde.reverse_posterior
    y = fn(torch.from_numpy(xT).to(DEV)).cpu().numpy()
    ref = g["unet_1x256x256/sampler_" + mode]
    got = y if mode == "sde" else sub3(y)
    e = relerr(got, ref)
    a = float(np.abs(np.asarray(got, dtype=np.float64) - ref).max())
    print("sampler 256x256 %s: %.3g rel, %.3g max-abs" % (mode, e, a))
    assert e < 2e-3, mode
    assert a < 1e-3, mode   # north_star: 1e-3 max-abs in fp32 on fixed noise (measured 6e-5)


def test_nafnet_512_and_T200_vs_reference_golden(golden):
    """Refusion ConditionalNAFNet at 1x3x512x512 (BASELINE configs[3] image size) and the T=200 / max_sigma 50 sampler."""
    g = golden.fullres
    m = refusion_net()
    lq, xT = O.synth_inputs(1234, 1, 512, 512, max_sigma=50)
    y = m(torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV), 60).cpu().numpy()
    scale = np.abs(g["naf_1x512x512/t60_sub3"]).max()
    e1 = float(np.abs(sub3(y) - g["naf_1x512x512/t60_sub3"]).max() / scale)
    e2 = float(np.abs(y[:, :, -64:, -64:] - g["naf_1x512x512/t60_corner"]).max() / scale)
    print("nafnet 512x512: %.3g %.3g" % (e1, e2))
    assert e1 < 1e-4 and e2 < 1e-4
    lqs, xTs = O.synth_inputs(1234, 1, 32, 32, max_sigma=50)
    sde = P.IRSDE(50, 200, "cosine", 0.005, device=DEV)
    sde.set_model(m)
    sde.set_mu(torch.from_numpy(lqs).to(DEV))
    sde.injected_noise = torch.from_numpy(O.synth_noise(7, 200, (1, 3, 32, 32))).to(DEV)
    for mode in ("sde", "posterior"):
        fn = sde.reverse_sde if mode == "sde" else sde.reverse_posterior
        e = relerr(fn(torch.from_numpy(xTs).to(DEV)).cpu().numpy(), g["naf_1x32x32_T200/" + mode])
        print("nafnet T=200 %s: %.3g" % (mode, e))
        assert e < 2e-3, mode


def _latent_256_models():
    um = P.latent.UNet(in_ch=3, out_ch=3, ch=64, ch_mult=[1, 2, 4], embed_dim=4)
    up = O.latent_unet_synth_params(seed=0, in_ch=3, out_ch=3, ch=64, ch_mult=(1, 2, 4), embed_dim=4)
    um.load_state_dict({k: torch.from_numpy(v) for k, v in up.items()}, strict=True)
    bm = P.latent_bokeh.ConditionalNAFNet(img_channel=4, width=64, enc_blk_nums=[1, 1, 1, 28], middle_blk_num=1, dec_blk_nums=[1, 1, 1, 1])
    bp = O.naf_synth_params(seed=3, img_channel=4, width=64, middle_blk_num=1, enc_blk_nums=(1, 1, 1, 28), dec_blk_nums=(1, 1, 1, 1), lens=True)
    bm.load_state_dict({k: torch.from_numpy(v) for k, v in bp.items()}, strict=True)
    return um.to(DEV).eval(), bm.to(DEV).eval()


def _latent_256_run(golden, dtype, unet_fp16=False):
    from image_restoration_sde_amd import _lib
    g = golden.fullres
    um, bm = _latent_256_models()
    bm.set_compute_dtype(dtype)
    if unet_fp16:
        um.engine_flags = _lib.FLAG_FP16   # the latent UNet's convolutions on fp16 operands too (encode / decode once per image)
    lq, _ = O.synth_inputs(1234, 1, 256, 256)
    lat, hid = um.encode(torch.from_numpy(lq).to(DEV))
    assert tuple(lat.shape) == (1, 4, 64, 64)
    T = 100
    sde = P.IRSDE(50, T, "cosine", 0.005, device=DEV)
    sde.set_model(bm)
    sde.set_mu(lat)
    sde.injected_noise = torch.from_numpy(O.synth_noise(7, T, (1, 4, 64, 64))).to(DEV)
    noisy = lat + torch.from_numpy(g["latent_1x256x256/z0"]).to(DEV) * sde.max_sigma
    lens = g["latent_1x256x256/lens"]
    li = [torch.from_numpy(lens[:, i].copy()) for i in range(3)]
    x0 = sde.reverse_sde(noisy, lens_info=li)
    rec = um.decode(x0, hid)
    return lat.cpu().numpy(), x0.cpu().numpy(), rec.cpu().numpy()


def test_latent_pipeline_256_vs_reference_golden(golden):
    """BASELINE configs[4] shape: 1x3x256x256 -> latent 64x64x4 (latent-bokeh networks), T=100 reverse_sde in the latent
    with lens_info, decode with the hidden skips — vs the REAL reference modules."""
    g = golden.fullres
    lat, x0, rec = _latent_256_run(golden, "fp32")
    e_lat = relerr(lat, g["latent_1x256x256/latent"])
    e_x0 = relerr(x0, g["latent_1x256x256/latent_sde"])
    e_rec = relerr(sub3(rec), g["latent_1x256x256/out_sde_sub3"])
    print("latent 256: encode %.3g, latent sampler %.3g, decoded %.3g" % (e_lat, e_x0, e_rec))
    assert e_lat < 1e-4 and e_x0 < 2e-3 and e_rec < 2e-3


@pytest.mark.parametrize("dtype,tol", [("fp16", 2e-3), ("bf16", 1e-1)])
def test_latent_pipeline_256_reduced_precision(golden, dtype, tol):
    """configs[4] names fp16: the latent score network with IEEE fp16 conv operands (IRSDE_FLAG_FP16, incl. naf_chain_kernel) stays within
    2e-3 of the fp32 reference over the T=100 sampler (relative to max|ref|; measured 1.4e-4 in r04 — the bar was 2e-2 until r05, VERDICT r04 weak #1a);
    the bf16 mode, with 8 significand bits, within 1e-1 — and fp16 must be the closer one."""
    g = golden.fullres
    _, x0, rec = _latent_256_run(golden, dtype)
    e_x0 = relerr(x0, g["latent_1x256x256/latent_sde"])
    e_rec = relerr(sub3(rec), g["latent_1x256x256/out_sde_sub3"])
    print("latent 256 %s: latent sampler %.3g, decoded %.3g" % (dtype, e_x0, e_rec))
    assert np.isfinite(rec).all() and e_x0 < tol and e_rec < tol
    _CACHE["latent_err_" + dtype] = e_x0
    if "latent_err_fp16" in _CACHE and "latent_err_bf16" in _CACHE:
        assert _CACHE["latent_err_fp16"] < _CACHE["latent_err_bf16"]


def test_latent_pipeline_256_all_fp16(golden):
    """configs[4] with BOTH networks in the fp16 operand mode (bench.py's latent workload): the latent UNet's encode / decode convolutions on fp16 operands
    as well as the score network.  Same 2e-3 bar as the score-network-only mode on the sampled latent and the decoded image (r04 measured 1.5e-4 / 3e-4); the encoder alone within 2e-3."""
    g = golden.fullres
    lat, x0, rec = _latent_256_run(golden, "fp16", unet_fp16=True)
    e_lat = relerr(lat, g["latent_1x256x256/latent"])
    e_x0 = relerr(x0, g["latent_1x256x256/latent_sde"])
    e_rec = relerr(sub3(rec), g["latent_1x256x256/out_sde_sub3"])
    print("latent 256 all-fp16: encode %.3g, latent sampler %.3g, decoded %.3g" % (e_lat, e_x0, e_rec))
    assert np.isfinite(rec).all() and e_lat < 2e-3 and e_x0 < 2e-3 and e_rec < 2e-3


# ---------------------------------------------------------------------------------------------
# production plan: batch independence, reduced precision at 256x256
# ---------------------------------------------------------------------------------------------
def test_configs2_batch16_bf16_act_plan(golden):
    """r06 (VERDICT r05 weak #1b / next #1 ii): BASELINE configs[2] AS BENCHMARKED — B = 16, 256 x 256, bf16_act.  That plan differs from the B = 1 plan the
    other reduced-precision tests run: conv3x3_halo2_kernel (the 512-pixel x 128-channel kernel) takes the layers with >= 256 blocks, the fused 16-bit attention
    kernels run 16 images wide.  (a) irsde_plan_describe names both; (b) one evaluation of the batch, slot 0 = the golden's inputs, against the fp32 engine at the
    B = 1 tolerance (_ATTN16_TOL) and against the B = 1 bf16_act plan on the same image; (c) T = 100 reverse_ode of the batch (sde_utils.py:268-282), slot 0 against
    the REAL reference's ODE golden (fullres2.npz) at _ODE_T100_TOL."""
    m = _fresh_unet64("bf16_act")
    buf = ctypes.create_string_buffer(1 << 16)
    _lib.check(_lib.lib().irsde_plan_describe(m.engine().h, 16, 256, 256, buf, len(buf)))
    d = buf.value
    assert d.count(b"kernel=halo512") >= 8 and d.count(b"kernel=halo256") >= 8, d.decode()
    assert d.count(b"k,v projection (bf16 operands + storage)") == 5 and d.count(b"q projection (bf16 operands + storage)") == 5
    _lib.check(_lib.lib().irsde_plan_describe(m.engine().h, 1, 256, 256, buf, len(buf)))
    n1 = buf.value.count(b"kernel=halo512")
    assert n1 < d.count(b"kernel=halo512")   # the B = 1 plan the other tests pin really is a different kernel mix
    lq, xT = O.synth_inputs(77, 16, 256, 256)
    lq1, xT1 = O.synth_inputs(1234, 1, 256, 256)
    lq[0], xT[0] = lq1[0], xT1[0]
    x, c = torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV)
    y16 = m(x, c, 50).cpu().numpy()
    y32 = unet64()(x[0:1], c[0:1], 50).cpu().numpy()
    y1 = m(x[0:1], c[0:1], 50).cpu().numpy()
    e32, e1, eref = relerr(y16[0:1], y32), relerr(y16[0:1], y1), relerr(y16[0:1], golden.fullres["unet_1x256x256/t50"])
    print("configs[2] plan (B=16 bf16_act), one evaluation, slot 0: vs fp32 engine %.3g, vs the B=1 bf16_act plan %.3g, vs the reference %.3g; halo512 rows %d (B=1: %d)"
          % (e32, e1, eref, d.count(b"kernel=halo512"), n1))
    assert np.isfinite(y16).all() and 0 < e32 < _ATTN16_TOL["bf16_act"] and eref < _ATTN16_TOL["bf16_act"] and e1 < _ATTN16_TOL["bf16_act"]
    sde = P.IRSDE(10, 100, "cosine", 0.005, device=DEV)
    sde.set_model(m)
    sde.set_mu(c)
    y = sde.reverse_ode(x).cpu().numpy()
    ref = golden.fullres2["unet_1x256x256/sampler_ode"]
    e = relerr(y[0:1], ref)
    print("configs[2] plan (B=16 bf16_act), T=100 reverse_ode, slot 0 vs the reference golden: %.3g (max-abs %.3g)" % (e, float(np.abs(y[0:1] - ref).max())))
    assert np.isfinite(y).all() and e < _ODE_T100_TOL["bf16_act"]


@pytest.mark.parametrize("prec", PRECS)
def test_batch16_256_equals_single_images(prec):
    """The plan at B=16 256x256 (tile-loop component GEMMs, 1x1 tile-loop path, 256-wide tiles) differs from the B=1 plan
    (64-row tiles, split-K): image b of the batch-16 evaluation must equal the single-image evaluation to fp32 noise."""
    m = unet64(prec)
    lq, xT = O.synth_inputs(77, 16, 256, 256)
    x, c = torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV)
    yb = m(x, c, 63).cpu().numpy()
    scale = np.abs(yb).max()
    worst = 0.0
    for b in range(16):
        y1 = m(x[b:b + 1], c[b:b + 1], 63).cpu().numpy()
        worst = max(worst, float(np.abs(y1 - yb[b:b + 1]).max() / scale))
    print("B=16 vs 16 x B=1 at 256x256: %.3g" % worst)
    assert worst < 5e-5
    # and the batch-16 plan against the reference golden through image 0 of the golden's inputs
    lq1, xT1 = O.synth_inputs(1234, 1, 256, 256)
    xx, cc = x.clone(), c.clone()
    xx[5], cc[5] = torch.from_numpy(xT1[0]).to(DEV), torch.from_numpy(lq1[0]).to(DEV)
    g = np.load(os.path.join(ROOT, "tests", "golden", "fullres.npz"))
    assert relerr(m(xx, cc, 50).cpu().numpy()[5:6], g["unet_1x256x256/t50"]) < 1e-4


def test_batch16_plan_sampler_256_vs_reference_golden(golden):
    """VERDICT r04 weak #1b: the BENCHMARKED plan (B = 16, 256 x 256, T = 100 reverse_sde on the captured step graph) pinned to the REAL reference at
    sampler level: slot 0 of the batch carries the golden's inputs and injected noise, the other 15 slots other images; slot 0 vs
    tests/golden/fullres.npz at north_star's 1e-3 max-abs (and 2e-3 relative)."""
    g = golden.fullres
    m = unet64()
    lq, xT = O.synth_inputs(77, 16, 256, 256)
    lq1, xT1 = O.synth_inputs(1234, 1, 256, 256)
    lq[0], xT[0] = lq1[0], xT1[0]
    z = O.synth_noise(7, 100, (1, 3, 256, 256))                       # the golden's noise for slot 0 ...
    zz = torch.from_numpy(O.synth_noise(8, 100, (16, 3, 256, 256)))    # ... other noise for the rest
    zz[:, 0] = torch.from_numpy(z[:, 0])
    sde = P.IRSDE(10, 100, "cosine", 0.005, device=DEV)
    sde.set_model(m)
    sde.set_mu(torch.from_numpy(lq).to(DEV))
    sde.injected_noise = zz.to(DEV)
    y = sde.reverse_sde(torch.from_numpy(xT).to(DEV)).cpu().numpy()
    ref = g["unet_1x256x256/sampler_sde"]
    e = relerr(y[0:1], ref)
    a = float(np.abs(np.asarray(y[0:1], dtype=np.float64) - ref).max())
    print("B=16 plan, slot 0, T=100 reverse_sde vs reference: %.3g rel, %.3g max-abs" % (e, a))
    assert np.isfinite(y).all() and e < 2e-3 and a < 1e-3


def test_fused_winograd_plan_vs_three_launch_plan():
    """The production plan at 2 x 256 x 256 runs the big feature maps on the fused Winograd kernel (csrc/wino_fused.hip);
    IRSDE_FLAG_NO_WINOGRAD_FUSED keeps them on the three-launch path.  Same arithmetic, different summation order."""
    m = unet64()
    buf = ctypes.create_string_buffer(1 << 16)
    _lib.check(_lib.lib().irsde_plan_describe(m.engine().h, 2, 256, 256, buf, len(buf)))
    assert b"winograd F4 fused" in buf.value
    params = O.synth_params(seed=0, nf=64, depth=4)
    m3 = P.ConditionalUNet(3, 3, 64, depth=4)
    m3.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m3.engine_flags = _lib.FLAG_NO_WINOGRAD_FUSED
    m3 = m3.to(DEV).eval()
    _lib.check(_lib.lib().irsde_plan_describe(m3.engine().h, 2, 256, 256, buf, len(buf)))
    assert b"winograd F4 fused" not in buf.value and b"winograd F4 gemm" in buf.value
    lq, xT = O.synth_inputs(31, 2, 256, 256)
    x, c = torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV)
    ya, yb = m(x, c, 42).cpu().numpy(), m3(x, c, 42).cpu().numpy()
    e = relerr(ya, yb)
    print("fused vs three-launch Winograd plan, 2x256x256: %.3g" % e)
    assert 0 < e < 5e-5


def test_fused_attention_plan_vs_qkv_tensor_plan():
    """Default fp32 plan: LinearAttention's k / v projection, softmax over the pixels and context run in one kernel (no k / v
    tensor in HBM); IRSDE_FLAG_NO_FUSED_ATTN keeps the to_qkv convolution + q|k|v tensor.  Same arithmetic up to summation
    order (odd sizes: 2 x 136 x 200 -> N = 27200 pixels at level 0, ragged chunk tails; attention at C = 64 .. 1024)."""
    m = unet64()
    buf = ctypes.create_string_buffer(1 << 16)
    _lib.check(_lib.lib().irsde_plan_describe(m.engine().h, 2, 136, 200, buf, len(buf)))
    assert b"k,v projection" in buf.value and b"+ context (fused)" in buf.value
    params = O.synth_params(seed=0, nf=64, depth=4)
    m3 = P.ConditionalUNet(3, 3, 64, depth=4)
    m3.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m3.engine_flags = _lib.FLAG_NO_FUSED_ATTN
    m3 = m3.to(DEV).eval()
    _lib.check(_lib.lib().irsde_plan_describe(m3.engine().h, 2, 136, 200, buf, len(buf)))
    assert b"(fused)" not in buf.value.replace(b"winograd F4 fused", b"")
    lq, xT = O.synth_inputs(32, 2, 136, 200)
    x, c = torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV)
    ya, yb = m(x, c, 17).cpu().numpy(), m3(x, c, 17).cpu().numpy()
    e = relerr(ya, yb)
    print("fused attention vs qkv-tensor plan, 2x136x200: %.3g" % e)
    assert 0 < e < 5e-5


# measured (r05, one evaluation at 2 x 136 x 200, relative to the fp32 engine): see profiles/r05_notes.md section 6
_ATTN16_TOL = {"bf16_act": 3e-2, "bf16": 3e-2, "fp16": 4e-3}


@pytest.mark.parametrize("dtype", ["bf16_act", "bf16", "fp16"])
def test_fused_attention_16bit_plan_vs_qkv_tensor_plan(dtype):
    """r05 (ABI 106): the reduced-precision modes run LinearAttention (C = 64 / 128 / 256) on the two fused kernels with bf16 / fp16 projection
    operands (bf16_act: bf16 tensors in and out) instead of to_qkv conv -> q|k|v tensor -> attention -> to_out conv.  Both plans round the same
    operands to 16 bits; the fused one no longer rounds q | k | v / the attention output to a stored tensor, so it must be at least as close
    to the fp32 engine as the tensor plan is (odd size: ragged tiles and chunk tails at every level)."""
    lq, xT = O.synth_inputs(32, 2, 136, 200)
    x, c = torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV)
    y32 = unet64()(x, c, 17).cpu().numpy()
    buf = ctypes.create_string_buffer(1 << 16)
    mf = _fresh_unet64(dtype)
    _lib.check(_lib.lib().irsde_plan_describe(mf.engine().h, 2, 136, 200, buf, len(buf)))
    tag = {"bf16_act": b"(bf16 operands + storage)", "bf16": b"(bf16 operands)", "fp16": b"(fp16 operands)"}[dtype]
    assert buf.value.count(b"k,v projection " + tag) == 5 and buf.value.count(b"q projection " + tag) == 5   # C = 64, 128, 256 down; 256, 128 up
    mu = _fresh_unet64(dtype)
    mu.engine_flags |= _lib.FLAG_NO_FUSED_ATTN
    _lib.check(_lib.lib().irsde_plan_describe(mu.engine().h, 2, 136, 200, buf, len(buf)))
    assert b"(fused)" not in buf.value
    yf, yu = mf(x, c, 17).cpu().numpy(), mu(x, c, 17).cpu().numpy()
    ef, eu, d = relerr(yf, y32), relerr(yu, y32), relerr(yf, yu)
    print("%s: fused attention plan vs fp32 %.3g, qkv-tensor plan vs fp32 %.3g, fused vs tensor plan %.3g" % (dtype, ef, eu, d))
    assert np.isfinite(yf).all() and 0 < ef < _ATTN16_TOL[dtype] and ef < 1.5 * eu and d > 0


def _fresh_unet64(dtype):
    params = O.synth_params(seed=0, nf=64, depth=4)
    mm = P.ConditionalUNet(3, 3, 64, depth=4)
    mm.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    mm = mm.to(DEV).eval()
    mm.set_compute_dtype(dtype)
    return mm


def test_ode_256_T100_vs_reference_golden(golden):
    """Full T=100 `reverse_ode` (sde_utils.py:268-282) at 1x3x256x256 in fp32 vs the REAL reference (fullres2.npz):
    the sampler BASELINE configs[2] names, at the shape it names."""
    m = unet64()
    lq, xT = O.synth_inputs(1234, 1, 256, 256)
    sde = P.IRSDE(10, 100, "cosine", 0.005, device=DEV)
    sde.set_model(m)
    sde.set_mu(torch.from_numpy(lq).to(DEV))
    y = sde.reverse_ode(torch.from_numpy(xT).to(DEV)).cpu().numpy()
    ref = golden.fullres2["unet_1x256x256/sampler_ode"]
    e = relerr(y, ref)
    print("sampler 256x256 ode T=100 fp32 vs reference: %.3g (max-abs %.3g)" % (e, float(np.abs(y - ref).max())))
    assert e < 2e-3 and float(np.abs(y - ref).max()) < 1e-3   # north_star: 1e-3 max-abs in fp32
    _CACHE["ode256_fp32"] = y


# Stated tolerances for the reduced-precision T=100 trajectory (relative to max|x0| of the fp32 result).  The reverse
# drift expands a per-evaluation perturbation ~200x over the high-theta steps t > 50, so these are ~200x the
# 20-step figures r02 quoted would suggest 4e-3 / 5e-4.  Measured on MI355X (profiles/r03_a_pytest_gpu.log, max|x0| = 28.8):
# bf16_act 3.2e-4 (max-abs 9.2e-3), bf16 3.1e-4 (8.9e-3), fp16 3.8e-5 (1.1e-3); the fp32 engine itself 1.8e-6 (5e-5).
_ODE_T100_TOL = {"bf16_act": 3e-3, "bf16": 3e-3, "fp16": 4e-4}


@pytest.mark.parametrize("dtype", ["bf16_act", "bf16", "fp16"])
def test_reduced_precision_ode_256_T100(golden, dtype):
    """BASELINE configs[2] as stated: `reverse_ode`, T=100, 256x256, reduced-precision operands — the WHOLE trajectory
    (r02 checked only the last 20 steps) vs the fp32 engine on the same weights and vs the reference's fp32 ODE golden."""
    lq, xT = O.synth_inputs(1234, 1, 256, 256)
    x, c = torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV)
    if "ode256_fp32" not in _CACHE:
        sde = P.IRSDE(10, 100, "cosine", 0.005, device=DEV)
        sde.set_model(unet64())
        sde.set_mu(c)
        _CACHE["ode256_fp32"] = sde.reverse_ode(x).cpu().numpy()
    sde = P.IRSDE(10, 100, "cosine", 0.005, device=DEV)
    sde.set_model(_fresh_unet64(dtype))
    sde.set_mu(c)
    y = sde.reverse_ode(x).cpu().numpy()
    e32 = relerr(y, _CACHE["ode256_fp32"])
    ref = golden.fullres2["unet_1x256x256/sampler_ode"]
    eref = relerr(y, ref)
    print("%s reverse_ode 256x256 T=100: vs fp32 engine %.3g, vs reference golden %.3g (max-abs %.3g)"
          % (dtype, e32, eref, float(np.abs(y - ref).max())))
    tol = _ODE_T100_TOL[dtype]
    assert np.isfinite(y).all() and 0 < e32 < tol and eref < tol
    _CACHE["ode_err_" + dtype] = e32
    if "ode_err_fp16" in _CACHE and "ode_err_bf16" in _CACHE:
        assert _CACHE["ode_err_fp16"] < _CACHE["ode_err_bf16"]   # 11 vs 8 significand bits


@pytest.mark.parametrize("prec", PRECS)
def test_unet_forward_512_vs_reference_golden_and_batch_plan(golden, prec):
    """ConditionalUNet.forward at 512x512 (north_star: "256x256 and 512x512 batches"): the single-image plan vs the REAL
    reference, then the 4 x 512 x 512 batch plan (1M-pixel level 0, 262 144 Winograd tiles per layer: other tile-loop /
    components-per-block choices than 16 x 256 x 256) vs four single-image evaluations and vs the golden through one slot."""
    g = golden.fullres2
    m = unet64(prec)
    lq1, xT1 = O.synth_inputs(1234, 1, 512, 512)
    x1, c1 = torch.from_numpy(xT1).to(DEV), torch.from_numpy(lq1).to(DEV)
    for t in (100, 23):
        y = m(x1, c1, t).cpu().numpy()
        scale = np.abs(g["unet_1x512x512/t%d_sub3" % t]).max()
        e1 = float(np.abs(sub3(y) - g["unet_1x512x512/t%d_sub3" % t]).max() / scale)
        e2 = float(np.abs(y[:, :, -48:, -48:] - g["unet_1x512x512/t%d_corner" % t]).max() / scale)
        print("unet 512x512 t=%d: %.3g %.3g" % (t, e1, e2))
        assert e1 < 1e-4 and e2 < 1e-4
    lq, xT = O.synth_inputs(78, 4, 512, 512)
    x, c = torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV)
    x[2], c[2] = x1[0], c1[0]
    yb = m(x, c, 23).cpu().numpy()
    scale = np.abs(yb).max()
    worst = 0.0
    for b in range(4):
        yy = m(x[b:b + 1], c[b:b + 1], 23).cpu().numpy()
        worst = max(worst, float(np.abs(yy - yb[b:b + 1]).max() / scale))
    print("B=4 vs 4 x B=1 at 512x512: %.3g" % worst)
    assert worst < 5e-5
    assert float(np.abs(sub3(yb[2:3]) - g["unet_1x512x512/t23_sub3"]).max() / np.abs(g["unet_1x512x512/t23_sub3"]).max()) < 1e-4


def test_unet_batch16_512_sliced_fused_layers_vs_reference_golden(golden):
    """r05: at 16 x 512 x 512 the 128-channel level-0 tensors pass 2 GiB — beyond the 32-bit buffer offsets of the fused Winograd kernels.  The plan
    now launches those layers on 2 slices of the batch (engine_plan.hip: push_wino_fused) instead of falling back to the three-launch path: the plan
    says so, image 11 of the batch equals the single-image evaluation, and image 3 (the golden's inputs) the REAL reference."""
    g = golden.fullres2
    m = unet64()
    buf = ctypes.create_string_buffer(1 << 16)
    _lib.check(_lib.lib().irsde_plan_describe(m.engine().h, 16, 512, 512, buf, len(buf)))
    assert b"batch slices)" in buf.value, buf.value[-800:]
    lq, xT = O.synth_inputs(79, 16, 512, 512)
    lq1, xT1 = O.synth_inputs(1234, 1, 512, 512)
    lq[3], xT[3] = lq1[0], xT1[0]
    x, c = torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV)
    yb = m(x, c, 23).cpu().numpy()
    assert np.isfinite(yb).all()
    y1 = m(x[11:12], c[11:12], 23).cpu().numpy()
    e = float(np.abs(y1 - yb[11:12]).max() / np.abs(yb).max())
    ref = g["unet_1x512x512/t23_sub3"]
    eg = float(np.abs(sub3(yb[3:4]) - ref).max() / np.abs(ref).max())
    print("16 x 512 x 512 (sliced fused layers): image 11 vs single %.3g, image 3 vs reference %.3g" % (e, eg))
    assert e < 5e-5 and eg < 1e-4


@pytest.mark.parametrize("prec", PRECS)
def test_nafnet_batch8_512_equals_single_images(golden, prec):
    """BASELINE configs[3] plan: Refusion NAFNet at 8 x 512 x 512 vs eight single-image evaluations (deterministic pooled
    sums of the SCA branch are per image; the batch plan picks other GEMM tilings), and vs the reference golden through
    one slot of the batch."""
    g = golden.fullres
    m = refusion_net(prec)
    lq1, xT1 = O.synth_inputs(1234, 1, 512, 512, max_sigma=50)
    lq, xT = O.synth_inputs(79, 8, 512, 512, max_sigma=50)
    x, c = torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV)
    x[6], c[6] = torch.from_numpy(xT1[0]).to(DEV), torch.from_numpy(lq1[0]).to(DEV)
    yb = m(x, c, 60).cpu().numpy()
    scale = np.abs(yb).max()
    worst = 0.0
    for b in range(8):
        yy = m(x[b:b + 1], c[b:b + 1], 60).cpu().numpy()
        worst = max(worst, float(np.abs(yy - yb[b:b + 1]).max() / scale))
    print("NAFNet B=8 vs 8 x B=1 at 512x512: %.3g" % worst)
    assert worst < 5e-5
    assert float(np.abs(sub3(yb[6:7]) - g["naf_1x512x512/t60_sub3"]).max() / np.abs(g["naf_1x512x512/t60_sub3"]).max()) < 1e-4


# ---------------------------------------------------------------------------------------------
# corner semantics
# ---------------------------------------------------------------------------------------------
def test_T0_runs_no_step():
    """reverse_*(x, T=0): the reference's range(1, 1) is empty and the clone of x comes back; the C ABI does the same."""
    m = unet64()
    lq, xT = O.synth_inputs(3, 1, 32, 32)
    x, c = torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV)
    sde = P.IRSDE(10, 100, "cosine", 0.005, device=DEV)
    sde.set_model(m)
    sde.set_mu(c)
    for fn in (sde.reverse_sde, sde.reverse_ode, sde.reverse_posterior):
        y = fn(x, T=0)
        assert torch.equal(y, x) and y.data_ptr() != x.data_ptr()
    sde.reverse_ode(x, T=1)  # engine + schedule exist now
    out = torch.full_like(x, float("nan"))
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.check(_lib.lib().irsde_sample(m.engine().h, 1, p(x), p(c), None, 0, 0, 1, 32, 32, 0, 0, p(out), None, 1))
    torch.cuda.synchronize()
    assert torch.equal(out, x)
    ds = P.DenoisingSDE(max_sigma=75, T=100, device=DEV)
    assert torch.equal(ds.reverse_ode(x, T=0), x)


def test_dsde_reverse_ode_ignores_x0_for_the_score():
    """DenoisingSDE.reverse_ode(xt, x0=GT) always uses the model's score (sde_utils.py:510-516: x0 only feeds the dumped
    real_score image); reverse_sde(xt, x0=GT) does replace it (:489-493)."""
    params = O.uncond_synth_params(seed=0, nf=32, depth=2)
    m = P.denoising_sde.ConditionalUNet(3, 3, 32, depth=2)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m = m.to(DEV).eval()
    lq, xT = O.synth_inputs(5, 1, 16, 16, max_sigma=25)
    x, gt = torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV)
    ds = P.DenoisingSDE(max_sigma=75, T=100, device=DEV)
    ds.set_model(m)
    a = ds.reverse_ode(x, T=6)
    b = ds.reverse_ode(x, x0=gt, T=6)
    assert torch.equal(a, b)
    ds.injected_noise = torch.from_numpy(O.synth_noise(7, 100, (1, 3, 16, 16))).to(DEV)
    c1 = ds.reverse_sde(x, T=6)
    c2 = ds.reverse_sde(x, x0=gt, T=6)
    assert not torch.equal(c1, c2)


def test_destroy_keeps_the_callers_device():
    """irsde_destroy runs from a Python finaliser: it must not change the thread's current device."""
    before = torch.cuda.current_device()
    m = P.ConditionalUNet(3, 3, 32, depth=2)
    params = O.synth_params(seed=0, nf=32, depth=2)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m = m.to(DEV).eval()
    lq, xT = O.synth_inputs(5, 1, 16, 16)
    m(torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV), 3)
    del m
    import gc
    gc.collect()
    assert torch.cuda.current_device() == before


# ---------------------------------------------------------------------------------------------
# the seam with reference-style objects
# ---------------------------------------------------------------------------------------------
class _ReferenceStyleIRSDE:
    """What the reference's IRSDE does with `self.model` (sde_utils.py:184-223, 252-299), restated with plain torch ops on
    whatever device the tensors live on: the foreign `sde` object of the seam.  The product's
    ConditionalUNet is driven through nothing but `model(x, mu, t)`."""

    def __init__(self, prod_sde, noise):
        s = prod_sde
        self.T, self.dt, self.max_sigma = s.T, s.dt, s.max_sigma
        self.thetas, self.sigmas, self.thetas_cumsum, self.sigma_bars = s.thetas, s.sigmas, s.thetas_cumsum, s.sigma_bars
        self.noise = noise
        self.calls = 0

    def set_mu(self, mu):
        self.mu = mu

    def set_model(self, model):
        self.model = model

    def _score(self, x, t):
        self.calls += 1
        return -self.model(x, self.mu, t) / self.sigma_bars[t]

    def reverse_sde(self, xt, T=-1, save_states=False, **kw):
        import math
        x = xt.clone()
        for t in reversed(range(1, self.T + 1)):
            drift = self.thetas[t] * (self.mu - x) - self.sigmas[t] ** 2 * self._score(x, t)
            x = x - drift * self.dt - self.sigmas[t] * (self.noise[t] * math.sqrt(self.dt))
        return x


def test_product_unet_inside_a_reference_style_sde(golden):
    """Seam direction 1: a FOREIGN sde object (reference-style per-step loop in torch) calling the product ConditionalUNet
    as a plain nn.Module reproduces the REAL reference's reverse_sde output (sampler.npz was written by the reference)."""
    g = golden.sampler
    tag = "nf64d4_1x32x32_T100"
    nf, depth, B, H, W, T = (int(v) for v in g[tag + "/cfg"])
    m = unet64()
    lq, xT = O.synth_inputs(1234, B, H, W)
    z = torch.from_numpy(O.synth_noise(7, T, (B, 3, H, W))).to(DEV)
    prod = P.IRSDE(10, T, "cosine", 0.005, device=DEV)
    foreign = _ReferenceStyleIRSDE(prod, z)
    foreign.set_model(torch.nn.DataParallel(m, device_ids=[0]) if False else m)
    foreign.set_mu(torch.from_numpy(lq).to(DEV))
    with torch.no_grad():
        out = foreign.reverse_sde(torch.from_numpy(xT).to(DEV)).cpu().numpy()
    assert foreign.calls == T
    assert relerr(out, g[tag + "/sde"]) < 2e-3


def test_product_sde_with_a_foreign_torch_model():
    """Seam direction 2: the product IRSDE driving a score model that is a plain PyTorch nn.Module (not ours): per-step
    model calls + the fused HIP update kernel; equals the reference-style loop around the same module."""
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(6, 16, 3, padding=1), torch.nn.SiLU(), torch.nn.Conv2d(16, 3, 3, padding=1)).to(DEV).eval()

    class Foreign(torch.nn.Module):
        def forward(self, x, mu, t, **kw):
            return net(torch.cat([x - mu, mu], 1)) * (0.1 + 0.001 * float(t))
    fm = Foreign()
    T = 10
    lq, xT = O.synth_inputs(8, 2, 20, 24)
    z = torch.from_numpy(O.synth_noise(7, T, (2, 3, 20, 24))).to(DEV)
    x, mu = torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV)
    prod = P.IRSDE(10, T, "cosine", 0.005, device=DEV)
    prod.set_model(fm)
    prod.set_mu(mu)
    prod.injected_noise = z
    with torch.no_grad():
        a = prod.reverse_sde(x).cpu().numpy()
        foreign = _ReferenceStyleIRSDE(prod, z)
        foreign.set_model(fm)
        foreign.set_mu(mu)
        b = foreign.reverse_sde(x).cpu().numpy()
    assert relerr(a, b) < 1e-5


@pytest.mark.skipif(not os.path.isdir("/root/reference/codes"), reason="the reference tree exists only in the build container")
def test_product_classes_inside_the_real_reference_wrapper():
    """Cross-seam test against the reference's OWN code (runs wherever a GPU and /root/reference are both present): the
    reference's `IRSDE` drives the product ConditionalUNet, and the product `IRSDE` is handed to a reference-style
    DenoisingModel.test; both must agree with the product's fused path."""
    sys.path.insert(0, ROOT)
    from oracle import gen_golden as G
    sde_utils, _RefUNet = G.load_reference("/root/reference")
    m = unet64()
    T = 6
    lq, xT = O.synth_inputs(2, 1, 32, 32)
    x, mu = torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV)
    ref_sde = sde_utils.IRSDE(max_sigma=10, T=T, schedule="cosine", eps=0.005, device=DEV)
    ref_sde.set_model(m)
    ref_sde.set_mu(mu)
    with torch.no_grad():
        a = ref_sde.reverse_ode(x).cpu().numpy()
    prod = P.IRSDE(10, T, "cosine", 0.005, device=DEV)
    prod.set_model(m)
    prod.set_mu(mu)
    b = prod.reverse_ode(x).cpu().numpy()
    assert relerr(b, a) < 1e-4


# ---------------------------------------------------------------------------------------------
# N > 1 launch path and the dataset evaluation driver
# ---------------------------------------------------------------------------------------------
def _run_bench(args, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` (no torchrun around it) must spawn its own two ranks and print one JSON line.  On a
    1-GPU box the two ranks share the GPU (IRSDE_BENCH_OVERSUBSCRIBE=1: gloo gather through the host, a test hook); with
    >= 2 GPUs this is the real thing over RCCL."""
    two = torch.cuda.device_count() >= 2
    res = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "2", "--size", "64", "--T", "6", "--no-cpu-baseline"],
                     None if two else {"IRSDE_BENCH_OVERSUBSCRIBE": "1"})
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 4 and res["value"] > 0 and res["scaling"] == "weak"
    assert ("RCCL" in res["config"]["parallelism"]) == two
    r = res["roofline"]
    assert 0 < r["frac"] <= 1 and 0 < r["mfma_kernel_frac"] <= 1 and r["frac"] <= r["mfma_kernel_frac"] + 1e-9


def test_bench_strong_scaling_ragged_two_ranks():
    """`--scaling strong`: ONE global batch split over the ranks (north_star: "image batches shard across the 8 GPUs").  5 images
    over 2 ranks = shards of 3 and 2 through `sample_shard` + the final gather; per-rank times are reported."""
    two = torch.cuda.device_count() >= 2
    res = _run_bench(["--gpus", "2", "--scaling", "strong", "--steps", "1", "--warmup", "1", "--batch", "5", "--size", "64", "--T", "6",
                      "--no-cpu-baseline"], None if two else {"IRSDE_BENCH_OVERSUBSCRIBE": "1"})
    assert res["n_gpus"] == 2 and res["scaling"] == "strong" and res["config"]["global_batch"] == 5 and res["value"] > 0
    assert 0 < res["rank_ms_per_step"]["min"] <= res["rank_ms_per_step"]["max"]
    assert "secondary" not in res


def test_bench_rejects_mismatched_world():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "does not match WORLD_SIZE" in (r.stdout + r.stderr)


def test_bench_single_gpu_line_small():
    """The default code path of bench.py (N=1) at a small shape: value on graph replay, roofline from the untimed pass."""
    res = _run_bench(["--steps", "1", "--warmup", "1", "--batch", "2", "--size", "64", "--T", "8", "--no-cpu-baseline"])
    assert res["n_gpus"] == 1 and res["unit"] == "images/s" and res["vs_baseline"] is None and res["dtype"] == "f32"
    r = res["roofline"]
    assert r["bound"] == "mfma" and 0 < r["frac"] <= 1 and r["unit"] == "TFLOP/s" and r["algorithmic_equiv_TFLOPs"] > 0
    assert abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-3   # (the compact line rounds both to 4 significant digits)


def _shard_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", device_id=dev)
    try:
        params = O.synth_params(seed=0, nf=32, depth=2)
        m = P.ConditionalUNet(3, 3, 32, depth=2)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
        m = m.to(dev).eval()
        lq, xT = O.synth_inputs(4, 5, 24, 20)
        sde = P.IRSDE(10, 8, "cosine", 0.005, device=dev)
        sde.set_model(m)
        sde.seed = 11
        out = P.sample_sharded(sde, "sde", torch.from_numpy(xT).to(dev), torch.from_numpy(lq).to(dev))
        q.put((rank, out.cpu().numpy()))
    finally:
        dist.destroy_process_group()


def test_sample_sharded_one_rank_over_rccl():
    """The RCCL branch on the hardware a 1-GPU box has: a process group of ONE rank with backend "nccl" (= RCCL), `init_process_group(..., device_id=)`,
    `sample_sharded` -> `sample_shard` -> the device-side `all_gather` of `gather_batch`.  Result == the plain single-GPU call, bit for bit."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_shard_worker, args=(0, 1, port, q))
    pr.start()
    rank, got = q.get(timeout=600)
    pr.join(timeout=120)
    assert pr.exitcode == 0 and rank == 0
    params = O.synth_params(seed=0, nf=32, depth=2)
    m = P.ConditionalUNet(3, 3, 32, depth=2)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m = m.to(DEV).eval()
    lq, xT = O.synth_inputs(4, 5, 24, 20)
    sde = P.IRSDE(10, 8, "cosine", 0.005, device=DEV)
    sde.set_model(m)
    sde.seed = 11
    sde.set_mu(torch.from_numpy(lq).to(DEV))
    ref = sde.reverse_sde(torch.from_numpy(xT).to(DEV)).cpu().numpy()
    assert got.shape == ref.shape and np.array_equal(got, ref)


def test_bench_one_rank_over_rccl():
    """bench.py's N > 1 code path (RCCL group, barrier fences, gather_batch, all_reduce'd phase agreement, max over ranks) with a group of one rank."""
    res = _run_bench(["--steps", "1", "--warmup", "1", "--batch", "3", "--size", "64", "--T", "6", "--no-cpu-baseline"], {"IRSDE_BENCH_FORCE_RCCL": "1"})
    assert res["n_gpus"] == 1 and res["value"] > 0 and res["config"]["global_batch"] == 3
    assert 0 < res["rank_ms_per_step"]["min"] <= res["rank_ms_per_step"]["max"]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_sample_sharded_two_real_gpus():
    """2-rank `sample_sharded` on real devices over RCCL: ragged split (5 images), Philox noise keyed by the global image
    index => every rank ends with exactly the single-GPU result."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    params = O.synth_params(seed=0, nf=32, depth=2)
    m = P.ConditionalUNet(3, 3, 32, depth=2)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m = m.to(DEV).eval()
    lq, xT = O.synth_inputs(4, 5, 24, 20)
    sde = P.IRSDE(10, 8, "cosine", 0.005, device=DEV)
    sde.set_model(m)
    sde.seed = 11
    sde.set_mu(torch.from_numpy(lq).to(DEV))
    want = sde.reverse_sde(torch.from_numpy(xT).to(DEV)).cpu().numpy()
    assert relerr(res[0], want) < 1e-5 and np.array_equal(res[0], res[1])


def test_eval_folder_on_a_synthetic_dataset(tmp_path):
    """tools/eval_folder.py (the loop of deraining/test.py:93-217) end to end on a synthetic LQ/GT folder with a
    reference-format checkpoint: restored PNGs are written, and the printed dataset averages equal the oracle's metric
    restatement applied to those PNGs."""
    import importlib.util
    from PIL import Image
    spec = importlib.util.spec_from_file_location("eval_folder", os.path.join(ROOT, "tools", "eval_folder.py"))
    ef = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ef)
    rs = np.random.RandomState(1)
    lq_dir, gt_dir, out_dir = tmp_path / "LQ", tmp_path / "GT", tmp_path / "res"
    lq_dir.mkdir()
    gt_dir.mkdir()
    names = []
    for i, (h, w) in enumerate([(40, 56), (40, 56), (33, 47), (40, 56), (40, 56)]):
        gt = rs.randint(0, 256, size=(h, w, 3), dtype=np.uint8)
        lq = np.clip(gt.astype(np.int32) + rs.randint(-20, 21, size=gt.shape), 0, 255).astype(np.uint8)
        Image.fromarray(gt).save(str(gt_dir / ("img%02d.png" % i)))
        Image.fromarray(lq).save(str(lq_dir / ("img%02d.png" % i)))
        names.append("img%02d" % i)
    params = O.synth_params(seed=0, nf=32, depth=2)
    ckpt = tmp_path / "G.pth"
    torch.save({"module." + k: torch.from_numpy(v) for k, v in params.items()}, ckpt)
    args = ["--lq", str(lq_dir), "--gt", str(gt_dir), "--weights", str(ckpt), "--nf", "32", "--depth", "2", "--T", "6",
            "--mode", "posterior", "--out", str(out_dir), "--batch", "2", "--seed", "3"]
    summary = ef.main(args)
    assert set(summary) == {"psnr", "ssim", "psnr_y", "ssim_y"}
    psnr, ssim = [], []
    for n in names:
        assert (out_dir / (n + ".png")).exists() and (out_dir / (n + "_LQ.png")).exists() and (out_dir / (n + "_HQ.png")).exists()
        o = np.asarray(Image.open(str(out_dir / (n + ".png"))), dtype=np.float64)[..., ::-1]       # BGR like cv2.imread
        gtv = np.asarray(Image.open(str(gt_dir / (n + ".png"))), dtype=np.float64)[..., ::-1]
        assert np.array_equal(np.asarray(Image.open(str(out_dir / (n + "_HQ.png")))), np.asarray(Image.open(str(gt_dir / (n + ".png")))))
        # the reference protocol: crop `crop_border or scale` = 4 px before the metrics (deraining/test.py:134, ir-sde.yml scale 4)
        psnr.append(O.calculate_psnr(o[4:-4, 4:-4], gtv[4:-4, 4:-4]))
        ssim.append(O.calculate_ssim(o[4:-4, 4:-4], gtv[4:-4, 4:-4]))
    assert abs(summary["psnr"] - float(np.mean(psnr))) < 1e-9
    assert abs(summary["ssim"] - float(np.mean(ssim))) < 1e-8
    # batching must not change a result: --batch 1 gives the same images
    out2 = tmp_path / "res1"
    s2 = ef.main(args[:-6] + ["--out", str(out2), "--batch", "1", "--seed", "3"])
    for n in names:
        a = np.asarray(Image.open(str(out_dir / (n + ".png"))), dtype=np.int32)
        b = np.asarray(Image.open(str(out2 / (n + ".png"))), dtype=np.int32)
        assert np.abs(a - b).max() <= 1
    assert abs(s2["psnr"] - summary["psnr"]) < 0.05
