"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C ABI of
libirsde_hip.so; the numpy oracle and the committed golden vectors (generated from the real reference)
are the checkers.  /root/reference is never read here.

Tolerances (floating point path, fp32 arithmetic on both sides):
  single convolution vs float64 oracle conv ........ 2e-5 relative to max|ref|  (direct and Winograd F(2x2,3x3));
                                                     5e-5 for Winograd F(4x4,3x3) (measured 5e-6 .. 1.4e-5)
  one UNet evaluation vs reference golden .......... 1e-4 relative to max|ref|  (~200 fp32 layers)
  one reverse step (elementwise) vs golden ......... 2e-6 abs / 1e-5 rel
  full sampler vs reference golden ................. 2e-3 relative to max|ref|  (reverse drift expands
      perturbations ~200x; the reference's own fp32-vs-fp64 gap is of the same size)
"""
import ctypes
import os

import numpy as np
import pytest
import torch

import image_restoration_sde_amd as P
from image_restoration_sde_amd import _lib
from oracle import irsde_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def relerr(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


def make_model(nf, depth, flags=0, seed=0):
    m = P.ConditionalUNet(3, 3, nf, depth=depth)
    params = O.synth_params(seed=seed, nf=nf, depth=depth)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m.engine_flags = flags
    return m.to(DEV).eval(), params


_MODELS = {}


def model(nf, depth, flags=0):
    key = (nf, depth, flags)
    if key not in _MODELS:
        _MODELS[key] = make_model(nf, depth, flags)
    return _MODELS[key]


def test_native_library_is_loaded():
    """The HIP extension (in-tree .so) is what runs; there is no eager/PyTorch fallback."""
    assert torch.cuda.is_available()
    L = _lib.lib()
    assert L.irsde_version() == 107
    maps = open("/proc/self/maps").read()
    assert "libirsde_hip.so" in maps


# ---------------------------------------------------------------------------------------------
# kernel level: implicit-GEMM convolution (every shape class the network uses)
# ---------------------------------------------------------------------------------------------
def run_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res, naive=0, splits=1, film_bstride=0, L=None):
    """x0/x1/res: numpy NCHW.  Returns numpy NCHW.  L: the library (default: the product build; `probes_lib()` for superseded kernel generations)."""
    L = L or _lib.lib()
    to_nhwc = lambda a: torch.from_numpy(np.ascontiguousarray(a.transpose(0, 2, 3, 1))).to(DEV)
    d0 = to_nhwc(x0)
    d1 = to_nhwc(x1) if x1 is not None else None
    B, C0, Hin, Win = x0.shape
    C1 = x1.shape[1] if x1 is not None else 0
    Cout, _, KH, KW = w.shape
    Ho = ((Hin << in_shift) + 2 * pad - KH) // stride + 1
    Wo = ((Win << in_shift) + 2 * pad - KW) // stride + 1
    out = torch.full((B, Ho, Wo, Cout), float("nan"), device=DEV)
    dres = to_nhwc(res) if res is not None else None
    dfilm = torch.from_numpy(film).to(DEV) if film is not None else None
    wc = np.ascontiguousarray(w, dtype=np.float32)
    bc = np.ascontiguousarray(bias, dtype=np.float32) if bias is not None else None
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    torch.cuda.synchronize()
    _lib.check(L.irsde_debug_conv(p(d0), C0, p(d1), C1, B, Hin, Win, in_shift, wc.ctypes.data_as(ctypes.c_void_p), Cout,
                                  KH, KW, stride, pad, bc.ctypes.data_as(ctypes.c_void_p) if bc is not None else None,
                                  p(dfilm), film_bstride, silu, p(dres), p(out), naive, splits, None), L)
    return out.cpu().numpy().transpose(0, 3, 1, 2)


def probes_lib():
    """libirsde_hip_probes.so (`make PROBES=1`: superseded kernel generations and measurement twins, not part of the product library); the
    tests that compare kernel generations skip when it has not been built."""
    try:
        return _lib.probes_lib()
    except _lib.IrsdeLibraryError as ex:
        pytest.skip(str(ex))


def oracle_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res, film_bstride=0):
    x = x0 if x1 is None else np.concatenate([x0, x1], axis=1)
    x = x.astype(np.float64)
    if in_shift:
        x = O.upsample_nearest2(x)
    y = O.conv2d(x, w.astype(np.float64), None if bias is None else bias.astype(np.float64), stride=stride, pad=pad)
    if film is not None:
        Cout = w.shape[0]
        f = film.astype(np.float64)
        if film_bstride:
            sc, sh = f[:, :Cout, None, None], f[:, Cout:, None, None]
        else:
            sc, sh = f[0, :Cout].reshape(1, -1, 1, 1), f[0, Cout:].reshape(1, -1, 1, 1)
        y = y * (sc + 1) + sh
    if silu:
        y = O.silu(y)
    if res is not None:
        y = y + res
    return y


CONV_CASES = {
    # name: (B, C0, C1, H, W, Cout, K, stride, pad, in_shift, bias, film, silu, res)
    "3x3_64_64_film_silu": (2, 64, 0, 24, 20, 64, 3, 1, 1, 0, False, True, True, False),
    "3x3_64_64_silu_res": (2, 64, 0, 24, 20, 64, 3, 1, 1, 0, False, False, True, True),
    "3x3_concat_192_128": (1, 128, 64, 16, 16, 128, 3, 1, 1, 0, False, True, True, False),
    "1x1_concat_res_conv": (1, 128, 64, 16, 16, 128, 1, 1, 0, 0, False, False, False, False),
    "1x1_qkv_384": (2, 64, 0, 12, 12, 384, 1, 1, 0, 0, False, False, False, False),
    "1x1_to_out_bias": (2, 128, 0, 12, 12, 256, 1, 1, 0, 0, True, False, False, False),
    "1x1_narrow_64_bias": (2, 128, 0, 12, 12, 64, 1, 1, 0, 0, True, False, False, False),
    "4x4_s2_down": (2, 64, 0, 16, 24, 128, 4, 2, 1, 0, True, False, False, False),
    "3x3_upsample_fused": (2, 128, 0, 8, 12, 64, 3, 1, 1, 1, True, False, False, False),
    "3x3_final_cout3": (2, 64, 0, 24, 20, 3, 3, 1, 1, 0, True, False, False, False),
    "3x3_final_cout3_large": (1, 64, 0, 256, 256, 3, 3, 1, 1, 0, True, False, False, False),   # >= 65536 pixels: vector-pipe kernel
    "3x3_m_tail_odd": (1, 32, 0, 7, 9, 96, 3, 1, 1, 0, False, False, True, False),
    "3x3_deep_k_1536": (1, 1024, 512, 4, 4, 256, 3, 1, 1, 0, False, True, True, False),
    "3x3_wino_res_bias": (2, 256, 0, 12, 20, 256, 3, 1, 1, 0, True, False, True, True),
    "3x3_wino_upsample": (1, 256, 0, 6, 10, 256, 3, 1, 1, 1, True, False, False, False),
    # r05: shapes for the 512-pixel x 128-channel halo kernel (conv3x3_halo2_kernel): several 16 x 32 tiles with ragged right / bottom edges, a channel
    # count that is not a multiple of the 128-wide block (masked columns), both concat sources, the fused upsample
    "3x3_halo2_ragged_160": (2, 64, 32, 20, 70, 160, 3, 1, 1, 0, True, True, True, True),
    "3x3_halo2_upsample_128": (1, 64, 0, 17, 33, 128, 3, 1, 1, 1, True, False, False, False),
}
HALO2_CASES = ["3x3_concat_192_128", "3x3_deep_k_1536", "3x3_wino_res_bias", "3x3_wino_upsample", "3x3_halo2_ragged_160", "3x3_halo2_upsample_128"]


@pytest.mark.parametrize("name", list(CONV_CASES))
def test_conv_kernel(name):
    B, C0, C1, H, W, Cout, K, stride, pad, in_shift, has_bias, has_film, silu, has_res = CONV_CASES[name]
    rs = np.random.RandomState(hash(name) % 2 ** 31)
    x0 = rs.standard_normal((B, C0, H, W)).astype(np.float32)
    x1 = rs.standard_normal((B, C1, H, W)).astype(np.float32) if C1 else None
    w = (rs.standard_normal((Cout, C0 + C1, K, K)) / np.sqrt((C0 + C1) * K * K)).astype(np.float32)
    bias = rs.standard_normal(Cout).astype(np.float32) if has_bias else None
    film = (0.3 * rs.standard_normal((1, 2 * Cout))).astype(np.float32) if has_film else None
    Ho = ((H << in_shift) + 2 * pad - K) // stride + 1
    Wo = ((W << in_shift) + 2 * pad - K) // stride + 1
    res = rs.standard_normal((B, Cout, Ho, Wo)).astype(np.float32) if has_res else None
    ref = oracle_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res)
    got = run_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res)
    assert got.shape == ref.shape and np.isfinite(got).all()
    assert relerr(got, ref) < 2e-5, name
    # the VALU cross-check kernel and the forced split-K path agree with the oracle too
    assert relerr(run_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res, naive=1), ref) < 2e-5
    assert relerr(run_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res, splits=3), ref) < 2e-5
    if K == 3 and stride == 1 and pad == 1 and Cout % 4 == 0 and Ho % 2 == 0 and Wo % 2 == 0:
        # Winograd F(2x2,3x3) path (input transform -> 16 batched MFMA GEMMs -> output transform + epilogue)
        assert relerr(run_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res, naive=2), ref) < 2e-5
        # same with the component GEMMs on the batch-loop kernel (all 16 / 2 components per block)
        assert relerr(run_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res, naive=12), ref) < 2e-5
        assert relerr(run_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res, naive=22), ref) < 2e-5
    if K == 3 and stride == 1 and pad == 1 and Cout % 4 == 0 and Ho % 4 == 0 and Wo % 4 == 0:
        # Winograd F(4x4,3x3): 36 batched GEMMs, 4x fewer multiplies, larger transform constants
        assert relerr(run_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res, naive=3), ref) < 5e-5
        assert relerr(run_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res, naive=13), ref) < 5e-5
        assert relerr(run_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res, naive=23), ref) < 5e-5
        if C0 % 32 == 0 and C1 % 32 == 0 and Cout % 32 == 0:
            # the fused Winograd F(4x4,3x3) kernel (wino_fused.hip): transforms inside the GEMM kernel
            assert relerr(run_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res, naive=33), ref) < 5e-5
        if C0 % 32 == 0 and C1 % 32 == 0 and Cout % 64 == 0 and (C0 + C1) % 64 == 0:
            # r03: the 64-cout fused Winograd kernel (16 tiles x 64 couts per block, v_mfma_f32_16x16x4_f32)
            assert relerr(run_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res, naive=34), ref) < 5e-5


@pytest.mark.parametrize("shape", [(3, 32, 0, 36, 44, 96, 0), (2, 32, 64, 8, 12, 32, 1), (1, 64, 0, 64, 96, 64, 0), (5, 96, 32, 4, 4, 160, 0)])
def test_conv_wino_fused_edges(shape):
    """The fused Winograd kernel on ragged tile groups (tile rows / columns that do not fill a 4 x 8 group), concat sources
    the fused upsample, per-sample FiLM rows, bias + SiLU + residual together."""
    B, C0, C1, H, W, Cout, up = shape
    rs = np.random.RandomState(B * 1000 + H)
    x0 = rs.standard_normal((B, C0, H, W)).astype(np.float32)
    x1 = rs.standard_normal((B, C1, H, W)).astype(np.float32) if C1 else None
    w = (rs.standard_normal((Cout, C0 + C1, 3, 3)) / np.sqrt((C0 + C1) * 9)).astype(np.float32)
    bias = rs.standard_normal(Cout).astype(np.float32)
    film = (0.3 * rs.standard_normal((B, 2 * Cout))).astype(np.float32)
    res = rs.standard_normal((B, Cout, H << up, W << up)).astype(np.float32)
    ref = oracle_conv(x0, x1, w, bias, 1, 1, up, film, 1, res, film_bstride=2 * Cout)
    got = run_conv(x0, x1, w, bias, 1, 1, up, film, 1, res, naive=33, film_bstride=2 * Cout)
    assert got.shape == ref.shape and np.isfinite(got).all()
    assert relerr(got, ref) < 5e-5, shape
    # no epilogue at all
    ref = oracle_conv(x0, x1, w, None, 1, 1, up, None, 0, None)
    assert relerr(run_conv(x0, x1, w, None, 1, 1, up, None, 0, None, naive=33), ref) < 5e-5, shape


@pytest.mark.parametrize("shape", [(3, 64, 0, 36, 44, 64, 0), (2, 32, 32, 8, 12, 128, 1), (1, 128, 64, 20, 28, 128, 0), (5, 96, 32, 4, 4, 192, 0),
                                   (1, 256, 0, 16, 16, 64, 1), (2, 64, 0, 64, 64, 64, 0)])
def test_conv_wino_fused64_edges(shape):
    """r03: the 64-cout fused Winograd kernel (wino4_fused64_kernel) on ragged 4 x 4 tile groups, concat sources whose boundary
    falls on a 32-channel chunk, the fused upsample, per-sample FiLM rows, bias + SiLU + residual together, 2 .. 8 chunks."""
    B, C0, C1, H, W, Cout, up = shape
    rs = np.random.RandomState(B * 1000 + H + 7)
    x0 = rs.standard_normal((B, C0, H, W)).astype(np.float32)
    x1 = rs.standard_normal((B, C1, H, W)).astype(np.float32) if C1 else None
    w = (rs.standard_normal((Cout, C0 + C1, 3, 3)) / np.sqrt((C0 + C1) * 9)).astype(np.float32)
    bias = rs.standard_normal(Cout).astype(np.float32)
    film = (0.3 * rs.standard_normal((B, 2 * Cout))).astype(np.float32)
    res = rs.standard_normal((B, Cout, H << up, W << up)).astype(np.float32)
    ref = oracle_conv(x0, x1, w, bias, 1, 1, up, film, 1, res, film_bstride=2 * Cout)
    got = run_conv(x0, x1, w, bias, 1, 1, up, film, 1, res, naive=34, film_bstride=2 * Cout)
    assert got.shape == ref.shape and np.isfinite(got).all()
    assert relerr(got, ref) < 5e-5, shape
    ref = oracle_conv(x0, x1, w, None, 1, 1, up, None, 0, None)
    got = run_conv(x0, x1, w, None, 1, 1, up, None, 0, None, naive=34)
    assert relerr(got, ref) < 5e-5, shape
    # and bit-for-bit nothing but summation order away from the 32-cout kernel
    assert relerr(got, run_conv(x0, x1, w, None, 1, 1, up, None, 0, None, naive=33)) < 2e-5, shape
    # the cout-block-by-XCD block mapping (forced where legal: NB in {2, 4, 8}, tile groups divisible) is the same arithmetic per block: bit-exact
    assert np.array_equal(got, run_conv(x0, x1, w, None, 1, 1, up, None, 0, None, naive=36)), shape


def _persistent_rounds_case(shape):
    B, C0, C1, H, W, Cout, up = shape
    rs = np.random.RandomState(B * 131 + Cout)
    x0 = rs.standard_normal((B, C0, H, W)).astype(np.float32)
    x1 = rs.standard_normal((B, C1, H, W)).astype(np.float32) if C1 else None
    w = (rs.standard_normal((Cout, C0 + C1, 3, 3)) / np.sqrt((C0 + C1) * 9)).astype(np.float32)
    bias = rs.standard_normal(Cout).astype(np.float32)
    film = (0.3 * rs.standard_normal((B, 2 * Cout))).astype(np.float32)
    res = rs.standard_normal((B, Cout, H << up, W << up)).astype(np.float32)
    ref = oracle_conv(x0, x1, w, bias, 1, 1, up, film, 1, res, film_bstride=2 * Cout)
    return (x0, x1, w, bias, 1, 1, up, film, 1, res), ref, 2 * Cout


PERSISTENT_SHAPES = [(3, 64, 0, 96, 128, 128, 0), (6, 32, 32, 32, 32, 192, 1), (6, 64, 64, 96, 128, 64, 0)]


@pytest.mark.parametrize("shape", PERSISTENT_SHAPES)
def test_conv_wino_fused64_persistent_rounds(shape):
    """r04: wino4_fused64p_kernel (production) is one block per CU walking its tile groups (288 items each: more than the 256 CUs, ragged last
    round): the producers run on into the next tile group (concat source / upsample offsets rebuilt at the boundary), the weight ring
    prefetches across it, the output transform is lane-local.  Against the float64 oracle; the cout-block-by-XCD item map (36 / 37) and the
    fixed-grid launches (55 / 56) are the same arithmetic in the same order: bit-exact; the fp16-pair twin (35) within the f32 bar."""
    args, ref, fb = _persistent_rounds_case(shape)
    got = run_conv(*args, naive=34, film_bstride=fb)
    assert got.shape == ref.shape and np.isfinite(got).all()
    assert relerr(got, ref) < 5e-5, shape
    assert np.array_equal(got, run_conv(*args, naive=36, film_bstride=fb)), shape
    assert np.array_equal(got, run_conv(*args, naive=55, film_bstride=fb)), shape
    pair = run_conv(*args, naive=35, film_bstride=fb)
    assert relerr(pair, ref) < 5e-5, shape
    assert np.array_equal(pair, run_conv(*args, naive=37, film_bstride=fb)), shape
    assert np.array_equal(pair, run_conv(*args, naive=56, film_bstride=fb)), shape


@pytest.mark.parametrize("shape", PERSISTENT_SHAPES)
def test_conv_wino_fused64_kernel_generations(shape):
    """The superseded / measurement kernels of the fused Winograd family live in the PROBES build only (libirsde_hip_probes.so, r05): r03's
    one-block-per-tile-group kernel (38 / 39), the halo kernel (57 / 58: patches through LDS), the single-stream kernel (60 / 61), the tuning twins of
    the persistent kernel (51 / 53: no double-fetched ring units / interleaved patch-load issue; 52: 12-operation B^T; 50: every tuning bit).  Same
    arithmetic in the same order wherever the input transform is not re-associated: bit-identical to the PRODUCT library's production kernel."""
    PL = probes_lib()
    args, ref, fb = _persistent_rounds_case(shape)
    got = run_conv(*args, naive=34, film_bstride=fb)                      # product library, production kernel
    assert np.array_equal(got, run_conv(*args, naive=34, film_bstride=fb, L=PL)), shape   # the probes build runs the same production kernel
    old = run_conv(*args, naive=38, film_bstride=fb, L=PL)
    assert relerr(old, ref) < 5e-5 and relerr(got, old) < 2e-5, shape
    for nv in (57, 51, 53, 60):
        assert np.array_equal(got, run_conv(*args, naive=nv, film_bstride=fb, L=PL)), (shape, nv)
    for nv in (52, 50):
        assert relerr(run_conv(*args, naive=nv, film_bstride=fb, L=PL), ref) < 5e-5, (shape, nv)
    pair = run_conv(*args, naive=35, film_bstride=fb)
    assert relerr(pair, run_conv(*args, naive=39, film_bstride=fb, L=PL)) < 2e-5, shape
    for nv in (58, 61):
        assert np.array_equal(pair, run_conv(*args, naive=nv, film_bstride=fb, L=PL)), (shape, nv)
    # the product library refuses these selectors loudly instead of silently running something else
    with pytest.raises(P.IrsdeError, match="PROBES"):
        run_conv(*args, naive=38, film_bstride=fb)


FUSED64T_SHAPES = [(3, 64, 0, 36, 44, 64, 0), (2, 32, 32, 8, 12, 128, 1), (1, 128, 64, 20, 28, 128, 0), (5, 96, 32, 4, 4, 192, 0), (1, 256, 0, 16, 16, 64, 1),
                   (2, 64, 0, 64, 64, 64, 0), (3, 64, 0, 96, 128, 128, 0), (6, 32, 32, 32, 32, 192, 1), (6, 64, 64, 96, 128, 64, 0), (4, 64, 0, 16, 32, 512, 0)]


@pytest.mark.parametrize("shape", FUSED64T_SHAPES)
def test_conv_wino_fused64t(shape):
    """r06: wino4_fused64t_kernel (csrc/wino_fused_t.hip; module_util.py:108-122) — 32 tiles x 64 couts per work item, every weight fragment feeds two tile groups,
    the input transform inside the matrix waves (half patches + v_permlane32_swap), the second stage of the output transform across two waves through LDS.
    Ragged 4 x 8 tile groups (tile columns 11 / 3 / 7 / 1 of 8), concat sources on a 32-channel boundary, the fused upsample, per-sample FiLM rows, bias + SiLU +
    residual, 4 .. 16 chunks, more work items than CUs (persistent rounds) and fewer.  Against the float64 oracle at the F(4x4) bar, and — same operations in the
    same order — BIT-IDENTICAL to the production kernel wino4_fused64p_kernel (34); the cout-block-by-XCD item map (63) bit-identical to the default one."""
    B, C0, C1, H, W, Cout, up = shape
    rs = np.random.RandomState(B * 131 + Cout + H)
    x0 = rs.standard_normal((B, C0, H, W)).astype(np.float32)
    x1 = rs.standard_normal((B, C1, H, W)).astype(np.float32) if C1 else None
    w = (rs.standard_normal((Cout, C0 + C1, 3, 3)) / np.sqrt((C0 + C1) * 9)).astype(np.float32)
    bias = rs.standard_normal(Cout).astype(np.float32)
    film = (0.3 * rs.standard_normal((B, 2 * Cout))).astype(np.float32)
    res = rs.standard_normal((B, Cout, H << up, W << up)).astype(np.float32)
    fb = 2 * Cout
    for args in ((x0, x1, w, bias, 1, 1, up, film, 1, res), (x0, x1, w, None, 1, 1, up, None, 0, None), (x0, x1, w, bias, 1, 1, up, None, 0, res),
                 (x0, x1, w, None, 1, 1, up, film, 1, None)):
        kw = {"film_bstride": fb} if args[7] is not None else {}
        ref = oracle_conv(*args, **kw)
        got = run_conv(*args, naive=62, **kw)
        assert got.shape == ref.shape and np.isfinite(got).all()
        e = relerr(got, ref)
        prod = run_conv(*args, naive=34, **kw)
        same = np.array_equal(got, prod)
        print("fused64t %s silu=%d res=%d: vs oracle %.3g, bit-identical to wino4_fused64p_kernel: %s (max diff %.3g)"
              % (shape, args[8], args[9] is not None, e, same, float(np.abs(got - prod).max())))
        assert e < 5e-5, shape
        assert same, shape
        assert np.array_equal(got, run_conv(*args, naive=63, **kw)), shape


@pytest.mark.parametrize("shape", [(2, 128, 0, 32, 32, 128, 0), (1, 64, 64, 16, 16, 256, 1), (4, 64, 0, 16, 32, 512, 0), (3, 64, 0, 16, 16, 128, 0)])
def test_conv_wino_fused64_xcd_mapping(shape):
    """wino4_fused64_kernel with cout block = XCD % NB (NB = 2 / 4 / 8; the last shape has 3 x 16 tile groups... an odd count the mapping
    must refuse for NB = 2 -> falls back): f32 (36) and fp16-pair (37) instances against the oracle and bit-exact against the default mapping."""
    B, C0, C1, H, W, Cout, up = shape
    rs = np.random.RandomState(B * 77 + Cout)
    x0 = rs.standard_normal((B, C0, H, W)).astype(np.float32)
    x1 = rs.standard_normal((B, C1, H, W)).astype(np.float32) if C1 else None
    w = (rs.standard_normal((Cout, C0 + C1, 3, 3)) / np.sqrt((C0 + C1) * 9)).astype(np.float32)
    bias = rs.standard_normal(Cout).astype(np.float32)
    res = rs.standard_normal((B, Cout, H << up, W << up)).astype(np.float32)
    ref = oracle_conv(x0, x1, w, bias, 1, 1, up, None, 1, res)
    a = run_conv(x0, x1, w, bias, 1, 1, up, None, 1, res, naive=34)
    b = run_conv(x0, x1, w, bias, 1, 1, up, None, 1, res, naive=36)
    assert relerr(b, ref) < 5e-5 and np.array_equal(a, b), shape
    c = run_conv(x0, x1, w, bias, 1, 1, up, None, 1, res, naive=35)
    d = run_conv(x0, x1, w, bias, 1, 1, up, None, 1, res, naive=37)
    assert relerr(d, ref) < 5e-5 and np.array_equal(c, d), shape


def test_conv_per_sample_film():
    rs = np.random.RandomState(5)
    x0 = rs.standard_normal((3, 32, 10, 10)).astype(np.float32)
    w = (rs.standard_normal((64, 32, 3, 3)) / 17).astype(np.float32)
    film = (0.3 * rs.standard_normal((3, 128))).astype(np.float32)
    ref = oracle_conv(x0, None, w, None, 1, 1, 0, film, 1, None, film_bstride=128)
    got = run_conv(x0, None, w, None, 1, 1, 0, film, 1, None, film_bstride=128)
    assert relerr(got, ref) < 2e-5


@pytest.mark.parametrize("name", ["3x3_64_64_film_silu", "3x3_concat_192_128", "1x1_qkv_384", "4x4_s2_down", "3x3_upsample_fused",
                                  "3x3_final_cout3", "3x3_m_tail_odd", "3x3_deep_k_1536", "3x3_wino_res_bias"])
def test_conv_kernel_bf16(name):
    """IRSDE_FLAG_BF16 kernel (v_mfma_f32_32x32x16_bf16): same operands as the oracle once both round to bf16 (RNE),
    products are exact in fp32, so only the fp32 accumulation order differs."""
    B, C0, C1, H, W, Cout, K, stride, pad, in_shift, has_bias, has_film, silu, has_res = CONV_CASES[name]
    rs = np.random.RandomState(hash(name) % 2 ** 31)
    x0 = rs.standard_normal((B, C0, H, W)).astype(np.float32)
    x1 = rs.standard_normal((B, C1, H, W)).astype(np.float32) if C1 else None
    w = (rs.standard_normal((Cout, C0 + C1, K, K)) / np.sqrt((C0 + C1) * K * K)).astype(np.float32)
    bias = rs.standard_normal(Cout).astype(np.float32) if has_bias else None
    film = (0.3 * rs.standard_normal((1, 2 * Cout))).astype(np.float32) if has_film else None
    Ho = ((H << in_shift) + 2 * pad - K) // stride + 1
    Wo = ((W << in_shift) + 2 * pad - K) // stride + 1
    res = rs.standard_normal((B, Cout, Ho, Wo)).astype(np.float32) if has_res else None
    with O.bf16_convs():
        ref = oracle_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res)
    full = oracle_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res)
    got = run_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res, naive=4)
    assert got.shape == ref.shape and np.isfinite(got).all()
    assert relerr(got, ref) < 2e-5, name
    assert 1e-4 < relerr(got, full) < 2e-2, name  # and it really is the reduced-precision product
    assert relerr(run_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res, naive=161), ref) < 2e-5  # 128x128 tile
    if Cout % 256 == 0:
        assert relerr(run_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res, naive=160), ref) < 2e-5  # 256x256 tile


@pytest.mark.parametrize("name", ["3x3_64_64_film_silu", "3x3_concat_192_128", "1x1_qkv_384", "4x4_s2_down", "3x3_upsample_fused",
                                  "3x3_m_tail_odd", "3x3_deep_k_1536", "3x3_wino_res_bias"])
def test_conv_kernel_fp16(name):
    """IRSDE_FLAG_FP16 kernels (v_mfma_f32_32x32x16_f16; BASELINE configs[4] names fp16): same operands as the oracle once
    both round to IEEE binary16 (RNE); products are exact in fp32, so only the fp32 accumulation order differs.  fp16
    keeps 11 significand bits: 8x closer to the fp32 result than the bf16 mode."""
    B, C0, C1, H, W, Cout, K, stride, pad, in_shift, has_bias, has_film, silu, has_res = CONV_CASES[name]
    rs = np.random.RandomState(hash(name) % 2 ** 31)
    x0 = rs.standard_normal((B, C0, H, W)).astype(np.float32)
    x1 = rs.standard_normal((B, C1, H, W)).astype(np.float32) if C1 else None
    w = (rs.standard_normal((Cout, C0 + C1, K, K)) / np.sqrt((C0 + C1) * K * K)).astype(np.float32)
    bias = rs.standard_normal(Cout).astype(np.float32) if has_bias else None
    film = (0.3 * rs.standard_normal((1, 2 * Cout))).astype(np.float32) if has_film else None
    Ho = ((H << in_shift) + 2 * pad - K) // stride + 1
    Wo = ((W << in_shift) + 2 * pad - K) // stride + 1
    res = rs.standard_normal((B, Cout, Ho, Wo)).astype(np.float32) if has_res else None
    with O.f16_convs():
        ref = oracle_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res)
    full = oracle_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res)
    got = run_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res, naive=5)
    assert got.shape == ref.shape and np.isfinite(got).all()
    assert relerr(got, ref) < 2e-5, name
    assert 1e-5 < relerr(got, full) < 3e-3, name  # and it really is the fp16-operand product (bf16 would be ~8x further off)
    assert relerr(run_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res, naive=165), ref) < 2e-5  # generic 128 tile


def test_unet_fp16_mode(golden):
    """IRSDE_FLAG_FP16 on the whole network: follows the oracle's restatement of the mode (conv operands rounded to fp16,
    everything else full precision) to 3e-3 of max|ref| and the fp32 reference to 5e-3; the plan is the bf16 mode's (no
    Winograd); a 20-step reverse_ode stays within 1e-2 of the fp32 engine."""
    params = O.synth_params(seed=0, nf=64, depth=4)
    m = P.ConditionalUNet(3, 3, 64, depth=4)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m = m.to(DEV).eval()
    m.set_compute_dtype("fp16")
    assert m.engine_flags == _lib.FLAG_FP16
    lq, xT = O.synth_inputs(1234, 1, 64, 64)
    y = m(torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV), 50).cpu().numpy()
    with O.f16_convs():
        ref16 = O.unet_forward(params, xT, lq, 50, depth=4, dtype=np.float64)
    e_oracle, e_fp32 = relerr(y, ref16), relerr(y, golden.forward["nf64d4_1x64x64/t50"])
    print("fp16 forward: vs fp16 oracle %.3g, vs fp32 reference %.3g" % (e_oracle, e_fp32))
    assert e_oracle < 3e-3 and 1e-6 < e_fp32 < 5e-3
    buf = ctypes.create_string_buffer(1 << 16)
    _lib.check(_lib.lib().irsde_plan_describe(m.engine().h, 1, 64, 64, buf, len(buf)))
    assert b"conv(fp16)" in buf.value and b"winograd" not in buf.value and b"conv(bf16)" not in buf.value
    m32, _ = model(64, 4)
    outs = {}
    for name, mm in (("fp16", m), ("fp32", m32)):
        sde = P.IRSDE(10, 100, "cosine", 0.005, device=DEV)
        sde.set_model(mm)
        sde.set_mu(torch.from_numpy(lq).to(DEV))
        outs[name] = sde.reverse_ode(torch.from_numpy(xT).to(DEV), T=20).cpu().numpy()
    e_ode = relerr(outs["fp16"], outs["fp32"])
    print("fp16 reverse_ode T=20 vs fp32: %.3g" % e_ode)
    assert e_ode < 1e-2
    # the flag cannot be combined with bf16 activation storage
    cfg = _lib.Config(3, 3, 64, 4, 0, _lib.FLAG_FP16 | _lib.FLAG_BF16_ACT)
    h = ctypes.c_void_p()
    assert _lib.lib().irsde_create(ctypes.byref(cfg), ctypes.byref(h)) != 0


@pytest.mark.parametrize("name", ["3x3_64_64_film_silu", "3x3_64_64_silu_res", "3x3_concat_192_128", "1x1_qkv_384", "4x4_s2_down",
                                  "3x3_upsample_fused", "3x3_m_tail_odd", "3x3_deep_k_1536", "3x3_wino_res_bias"])
def test_conv_kernel_bf16_storage(name):
    """IRSDE_FLAG_BF16_ACT kernels: bf16 tensors in HBM on both sides (inputs, residual, output).  Against the oracle with
    the same roundings (operands, residual; fp32+ arithmetic; one final rounding of the stored result): one bf16 ulp
    (2^-8 relative to the value) is the most a different accumulation order can move a stored element."""
    B, C0, C1, H, W, Cout, K, stride, pad, in_shift, has_bias, has_film, silu, has_res = CONV_CASES[name]
    rs = np.random.RandomState(hash(name) % 2 ** 31)
    x0 = rs.standard_normal((B, C0, H, W)).astype(np.float32)
    x1 = rs.standard_normal((B, C1, H, W)).astype(np.float32) if C1 else None
    w = (rs.standard_normal((Cout, C0 + C1, K, K)) / np.sqrt((C0 + C1) * K * K)).astype(np.float32)
    bias = rs.standard_normal(Cout).astype(np.float32) if has_bias else None
    film = (0.3 * rs.standard_normal((1, 2 * Cout))).astype(np.float32) if has_film else None
    Ho = ((H << in_shift) + 2 * pad - K) // stride + 1
    Wo = ((W << in_shift) + 2 * pad - K) // stride + 1
    res = rs.standard_normal((B, Cout, Ho, Wo)).astype(np.float32) if has_res else None
    with O.bf16_convs():
        ref = oracle_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, None if res is None else O.round_bf16(res))
    ref_st = O.round_bf16(ref)
    for code in (204, 261) + ((260,) if Cout % 256 == 0 else ()):   # automatic (halo for 3x3) / generic 128 / generic 256 tile
        got = run_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res, naive=code)
        assert np.isfinite(got).all()
        assert np.array_equal(got, O.round_bf16(got))                       # what comes back was stored as bf16
        err = np.abs(got - ref_st)
        assert (err <= np.abs(ref_st) * 2.0 ** -7 + 1e-6).all(), (name, code, float(err.max()))
        assert (err > 0).mean() < 0.02, (name, code)                         # and almost every element is identical


@pytest.mark.parametrize("name", HALO2_CASES)
def test_conv_halo2_kernel(name):
    """r05: conv3x3_halo2_kernel (16 x 32 pixels x 128 channels per block, 4 x 2 MFMA tiles per wave, three taps per barrier) forced on small shapes
    (irsde_debug_conv 162 / 262 / 166) in its three instances — bf16 operands on fp32 tensors, bf16 tensors, fp16 operands — against the oracle with the
    same operand roundings, and against the 256-pixel kernel it replaces on the big feature maps (163 / 263 / 167)."""
    B, C0, C1, H, W, Cout, K, stride, pad, in_shift, has_bias, has_film, silu, has_res = CONV_CASES[name]
    rs = np.random.RandomState(hash(name) % 2 ** 31)
    x0 = rs.standard_normal((B, C0, H, W)).astype(np.float32)
    x1 = rs.standard_normal((B, C1, H, W)).astype(np.float32) if C1 else None
    w = (rs.standard_normal((Cout, C0 + C1, K, K)) / np.sqrt((C0 + C1) * K * K)).astype(np.float32)
    bias = rs.standard_normal(Cout).astype(np.float32) if has_bias else None
    film = (0.3 * rs.standard_normal((1, 2 * Cout))).astype(np.float32) if has_film else None
    Ho, Wo = H << in_shift, W << in_shift
    res = rs.standard_normal((B, Cout, Ho, Wo)).astype(np.float32) if has_res else None
    args = (x0, x1, w, bias, stride, pad, in_shift, film, silu)
    with O.bf16_convs():
        ref = oracle_conv(*args, res)
        ref_act = O.round_bf16(oracle_conv(*args, None if res is None else O.round_bf16(res)))
    with O.f16_convs():
        ref16 = oracle_conv(*args, res)
    got = run_conv(*args, res, naive=162)
    assert got.shape == ref.shape and np.isfinite(got).all()
    e_new, e_old = relerr(got, ref), relerr(run_conv(*args, res, naive=163), ref)
    assert e_new < 2e-5 and e_old < 2e-5, (name, e_new, e_old)
    e16 = relerr(run_conv(*args, res, naive=166), ref16)
    assert e16 < 2e-5, (name, e16)
    ga = run_conv(*args, res, naive=262)
    assert np.array_equal(ga, O.round_bf16(ga))
    err = np.abs(ga - ref_act)
    assert (err <= np.abs(ref_act) * 2.0 ** -7 + 1e-6).all() and (err > 0).mean() < 0.02, (name, float(err.max()))
    print("%s: halo2 bf16 %.3g (256-pixel kernel %.3g), fp16 %.3g, bf16 storage max %.3g" % (name, e_new, e_old, e16, float(err.max())))


@pytest.mark.parametrize("tag", ["nf64d4_1x32x32", "nf64d4_1x64x64", "nf64d4_2x40x56"])
def test_unet_forward_attention_sensitive_vs_reference(golden, tag):
    """r05: ConditionalUNet.forward against the REAL reference with attention-sensitive weights (tests/golden/forward_attn.npz; O.attn_sensitive_params: the network
    output moves by 0.4 of its maximum when the LinearAttention branches change — with the default synthetic weights it moves by 1e-6).  fp32 engine, all nine
    attention blocks (fused kernels at 64 / 128 / 256 channels, the q | k | v tensor path below), incl. the reflect-padded 40 x 56 case."""
    g = golden.forward_attn
    nf, depth, B, H, W, t = (int(v) for v in g[tag + "/cfg"])
    params = O.attn_sensitive_params(O.synth_params(seed=0, nf=nf, depth=depth), H, W, depth)
    m = P.ConditionalUNet(3, 3, nf, depth=depth)
    m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in params.items()}, strict=True)
    m = m.to(DEV).eval()
    lq, xT = O.synth_inputs(1234, B, H, W)
    y = m(torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV), t).cpu().numpy()
    e = relerr(y, g[tag + "/y"])
    print("attention-sensitive forward %s vs the reference: %.3g" % (tag, e))
    assert e < 1e-4


def test_unet_forward_attention_sensitive_256_vs_reference(golden):
    """r06 (VERDICT r05 weak #1a / next #1): the attention-sensitive forward of the REAL reference at the BENCHMARKED image size (tests/golden/forward_attn256.npz,
    oracle/gen_golden.py --only forward_attn256; module_util.py:150-178): N = 65 536 pixels at level 0.  B = 1 and, through slot 3 of a batch of 4, the multi-image
    grid of the fused kernels.  The fixture itself says how sensitive it is (the scaled to_out weights move the output by 0.43 of its maximum)."""
    g = golden.forward_attn256
    tag = "nf64d4_1x256x256"
    nf, depth, B, H, W, t = (int(v) for v in g[tag + "/cfg"])
    assert float(g[tag + "/moved_by"]) > 0.2
    params = O.attn_sensitive_params(O.synth_params(seed=0, nf=nf, depth=depth), H, W, depth)
    m = P.ConditionalUNet(3, 3, nf, depth=depth)
    m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in params.items()}, strict=True)
    m = m.to(DEV).eval()
    lq, xT = O.synth_inputs(1234, B, H, W)
    scale = float(g[tag + "/y_absmax"])

    def err(y):
        return max(float(np.abs(y[..., 1::3, 2::3] - g[tag + "/y_sub3"]).max()), float(np.abs(y[:, :, -48:, -48:] - g[tag + "/y_corner"]).max())) / scale
    e1 = err(m(torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV), t).cpu().numpy())
    lq4, xT4 = O.synth_inputs(99, 4, H, W)
    lq4[3], xT4[3] = lq[0], xT[0]
    e4 = err(m(torch.from_numpy(xT4).to(DEV), torch.from_numpy(lq4).to(DEV), t).cpu().numpy()[3:4])
    print("attention-sensitive forward 1x256x256 vs the reference: B=1 %.3g, slot 3 of B=4 %.3g" % (e1, e4))
    assert e1 < 1e-4 and e4 < 1e-4


_ATTN_BLOCK_LEVEL = {"downs.0.2.": 0, "downs.1.2.": 1, "downs.2.2.": 2, "ups.2.2.": 1, "ups.3.2.": 0}   # the fused blocks (C = 64 / 128 / 256) and their resolution level
_ATTN_BLOCK_LEVEL_DEEP = {"downs.3.2.": 3, "mid_attn.": 3, "ups.0.2.": 3, "ups.1.2.": 2}                # C = 512 / 1024: to_qkv conv + attention kernels + to_out conv (+ LayerNorm)


@pytest.mark.parametrize("dtype,size", [("fp32", (64, 64)), ("fp32", (72, 88)), ("bf16_act", (64, 64)), ("bf16", (64, 64)), ("fp16", (64, 64)), ("fp16", (72, 88)),
                                        ("fp32", (256, 256)), ("bf16_act", (256, 256))])
def test_fused_attention_block_vs_oracle(dtype, size):
    """r05: every fused LinearAttention block against the oracle, block by block — fp32 (the headline's kernels) against the float64 block at 5e-5 of the branch, the
    16-bit operand modes (ABI 106) against the oracle's restatement of their roundings (O.attn_block_fused16).  The block's input is read back from the engine (debug
    taps), the oracle computes the block from it in float64.  With the default synthetic weights
    the block's output is dominated by to_out's bias (the context carries v / N: O(1e-4) at 64 x 64), so a wrong attention core would move the result by 1e-6 —
    here to_out.0.weight is scaled by the level's pixel count, which makes the branch O(1) and every stage of the core visible in it.  72 x 88 (padded to 80 x 96):
    480 pixels at level 2 = 3.75 tiles of 128, chunk tails that are no multiple of a tile at every level (the masked paths of both kernels).
    r06 (VERDICT r05 weak #1a): 256 x 256 with B = 2 — the image size of the benchmarked plan: 512 tiles of 128 pixels per image at level 0, 16 tiles per chunk, more
    than one image in the grid — for the headline's fp32 kernels and for configs[2]'s bf16_act kernels."""
    nf, depth = 64, 4
    nb = 2 if size == (256, 256) else 1
    Hp, Wp = -(-size[0] // 16) * 16, -(-size[1] // 16) * 16
    params = dict(O.synth_params(seed=0, nf=nf, depth=depth))
    blocks = dict(_ATTN_BLOCK_LEVEL)
    if dtype == "fp32":
        blocks.update(_ATTN_BLOCK_LEVEL_DEEP)   # the q | k | v tensor path of the deep levels against the same oracle
    for pref, lvl in blocks.items():
        params[pref + "fn.fn.to_out.0.weight"] = params[pref + "fn.fn.to_out.0.weight"] * np.float32((Hp >> lvl) * (Wp >> lvl))
    m = P.ConditionalUNet(3, 3, nf, depth=depth)
    m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in params.items()}, strict=True)
    if dtype != "fp32":
        m.set_compute_dtype(dtype)
    m.engine_flags |= _lib.FLAG_KEEP_ACTIVATIONS
    m = m.to(DEV).eval()
    lq, xT = O.synth_inputs(1234, nb, size[0], size[1])
    m(torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV), 50)
    buf = ctypes.create_string_buffer(1 << 16)
    _lib.check(_lib.lib().irsde_plan_describe(m.engine().h, nb, size[0], size[1], buf, len(buf)))
    assert buf.value.count(b"+ context (fused)") == 5 and buf.value.count(b"+ residual (fused)") == 5
    p64 = {k: np.asarray(v, dtype=np.float64) for k, v in params.items()}
    for pref in blocks:
        xin = m.debug_tap("mid_block1" if pref == "mid_attn." else pref[:-3] + ".1").numpy().astype(np.float64)
        got = m.debug_tap(pref[:-1]).numpy().astype(np.float64)
        full = O.attn_block(p64, pref, xin)
        ref = full if dtype == "fp32" else O.attn_block_fused16(p64, pref, xin, f16=dtype == "fp16", store_bf16=dtype == "bf16_act")
        branch = float(np.abs(full - xin).max())
        e_ref, e_full = float(np.abs(got - ref).max()) / branch, float(np.abs(got - full).max()) / branch
        rms_ref, rms_full = float(np.sqrt(((got - ref) ** 2).mean())) / branch, float(np.sqrt(((got - full) ** 2).mean())) / branch
        print("%s %s C=%d: branch max %.3g; vs the oracle's restatement max %.3g rms %.3g, vs the unrounded block max %.3g rms %.3g"
              % (dtype, pref, xin.shape[1], branch, e_ref, rms_ref, e_full, rms_full))
        assert branch > 0.5, pref                      # the attention branch is what is being compared
        if dtype == "fp32":
            assert e_ref < 5e-5, (pref, e_ref)         # measured 5e-7
        elif dtype == "bf16_act":                      # + one final rounding of the stored result: one bf16 ulp on top of the bar below, and few elements differ at all
            err = np.abs(got - ref)
            assert (err <= np.abs(ref) * 2.0 ** -7 + 3e-3 * branch).all() and (err > 0).mean() < 0.25, (pref, float(err.max()), float((err > 0).mean()))
        else:
            # measured (r05 call T / tools/dbg_attn16_dump.py): bf16 max 1.1e-3 rms 8e-5 .. 1.1e-4, fp16 max 2e-4 rms 8e-6 .. 2.3e-5 — the maximum is set by single
            # tie flips of the attention output in front of the scaled to_out weights; leaving ANY of the restated roundings out of the oracle (other than exp(k) / v)
            # triples the rms.  The restatement must explain the kernel: closer to it than the unrounded block is, by 2x in rms
            assert e_ref < (6e-4 if dtype == "fp16" else 3e-3) and rms_ref < (4e-5 if dtype == "fp16" else 2e-4), (pref, e_ref, rms_ref)
            assert rms_ref < 0.5 * rms_full, (pref, rms_ref, rms_full)


# ---------------------------------------------------------------------------------------------
# network level
# ---------------------------------------------------------------------------------------------
def test_unet_bf16_act_mode(golden):
    """IRSDE_FLAG_BF16_ACT (bf16 conv operands + bf16 storage of every activation tensor): follows the oracle's restatement
    of the mode, stays close to the fp32 reference, and the ODE sampler stays close to the fp32 engine."""
    g = golden.forward
    tag = "nf64d4_1x64x64"
    nf, depth, B, H, W = (int(v) for v in g[tag + "/cfg"])
    m, params = make_model(nf, depth)
    m.set_compute_dtype("bf16_act")
    lq, xT = O.synth_inputs(1234, B, H, W)
    x, c = torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV)
    t = int(g[tag + "/ts"][1])
    y = m(x, c, t).cpu().numpy()
    with O.bf16_convs(store_bf16=True):
        ref = O.unet_forward(params, xT, lq, t, depth=depth, dtype=np.float64)
    e_oracle, e_fp32 = relerr(y, ref), relerr(y, g[tag + "/t%d" % t])
    print("bf16_act forward: vs oracle %.3g, vs fp32 reference %.3g" % (e_oracle, e_fp32))
    assert e_oracle < 3e-2
    assert 1e-4 < e_fp32 < 5e-2
    m32, _ = model(nf, depth)
    T = 20
    outs = {}
    for name, mm in (("bf16_act", m), ("fp32", m32)):
        sde = P.IRSDE(max_sigma=10, T=T, schedule="cosine", eps=0.005, device=DEV)
        sde.set_model(mm)
        sde.set_mu(c)
        outs[name] = sde.reverse_ode(x).cpu().numpy()
    e_ode = relerr(outs["bf16_act"], outs["fp32"])
    print("bf16_act reverse_ode T=20 vs fp32: %.3g" % e_ode)
    assert e_ode < 5e-2
    # debug taps read the bf16 tensors back correctly (first ResBlock vs the oracle's stored value)
    mk, _ = make_model(nf, depth, flags=_lib.FLAG_KEEP_ACTIVATIONS | _lib.FLAG_BF16 | _lib.FLAG_BF16_ACT)
    taps = {}
    with O.bf16_convs(store_bf16=True):
        O.unet_forward(params, xT, lq, t, depth=depth, dtype=np.float64, taps=taps)
    mk(x, c, t)
    for name in ("init_conv", "downs.0.0", "downs.0.2"):
        assert relerr(mk.debug_tap(name).numpy(), taps[name]) < 2e-2, name


def test_unet_bf16_act_batch_and_padding_properties():
    """bf16-storage mode on a reflect-padded odd size with per-image timesteps ([B] tensor -> per-sample FiLM rows in the
    halo / generic / split-K epilogues): image b of the batch agrees with the single-image call with scalar t, and the result
    stays close to fp32."""
    B, H, W = 3, 40, 56
    m, _ = make_model(64, 4)
    m.set_compute_dtype("bf16_act")
    m32, _ = model(64, 4)
    lq, xT = O.synth_inputs(77, B, H, W)
    x, c = torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV)
    ts = torch.tensor([5, 60, 99])
    yb = m(x, c, ts).cpu().numpy()
    assert np.isfinite(yb).all()
    for b in range(B):
        y1 = m(x[b:b + 1], c[b:b + 1], int(ts[b])).cpu().numpy()
        assert relerr(y1, yb[b:b + 1]) < 2e-2   # tile geometry (hence fp32 summation order + bf16 flips) differs with B
    y32 = m32(x, c, ts).cpu().numpy()
    assert relerr(yb, y32) < 5e-2


def test_unet_bf16_mode(golden):
    """BASELINE configs[2]: bf16 conv operands.  (a) the engine follows the oracle's restatement of the mode (operands
    rounded, everything else full precision) to 2e-2 of max|ref| — a value that sits on a bf16 rounding boundary may
    round the other way on the two sides; (b) it stays within 3e-2 of the fp32 reference golden; (c) deterministic ODE
    sampling in bf16 stays within 5e-2 of the fp32 engine over 20 steps."""
    g = golden.forward
    tag = "nf64d4_1x64x64"
    nf, depth, B, H, W = (int(v) for v in g[tag + "/cfg"])
    m, params = make_model(nf, depth)
    m.set_compute_dtype("bf16")
    lq, xT = O.synth_inputs(1234, B, H, W)
    x, c = torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV)
    t = int(g[tag + "/ts"][1])
    y = m(x, c, t).cpu().numpy()
    with O.bf16_convs():
        ref = O.unet_forward(params, xT, lq, t, depth=depth, dtype=np.float64)
    e_oracle, e_fp32 = relerr(y, ref), relerr(y, g[tag + "/t%d" % t])
    print("bf16 forward: vs bf16 oracle %.3g, vs fp32 reference %.3g" % (e_oracle, e_fp32))
    assert e_oracle < 2e-2
    assert 1e-4 < e_fp32 < 3e-2
    import ctypes as _c
    buf = _c.create_string_buffer(1 << 18)
    _lib.check(_lib.lib().irsde_plan_describe(m.engine().h, B, H, W, buf, len(buf)))
    assert b"conv(bf16)" in buf.value and b"winograd" not in buf.value
    # sampler: reverse_ode, bf16 vs fp32 engine (same weights)
    m32, _ = model(nf, depth)
    T = 20
    outs = {}
    for name, mm in (("bf16", m), ("fp32", m32)):
        sde = P.IRSDE(max_sigma=10, T=T, schedule="cosine", eps=0.005, device=DEV)
        sde.set_model(mm)
        sde.set_mu(c)
        outs[name] = sde.reverse_ode(x).cpu().numpy()
    e_ode = relerr(outs["bf16"], outs["fp32"])
    print("bf16 reverse_ode T=20 vs fp32: %.3g" % e_ode)
    assert e_ode < 5e-2


@pytest.mark.parametrize("tag", ["nf32d2_2x24x20", "nf64d4_1x64x64", "nf64d4_2x40x56"])
def test_unet_forward_vs_reference_golden(golden, tag):
    g = golden.forward
    nf, depth, B, H, W = (int(v) for v in g[tag + "/cfg"])
    m, _ = model(nf, depth)
    lq, xT = O.synth_inputs(1234, B, H, W)
    x, c = torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV)
    for t in g[tag + "/ts"]:
        y = m(x, c, int(t)).cpu().numpy()
        assert relerr(y, g[tag + "/t%d" % t]) < 1e-4, (tag, int(t))
    # tensor-valued time (tensor([t]) on device, DenoisingUNet_arch.py:87-88)
    t0 = int(g[tag + "/ts"][0])
    y = m(x, c, torch.tensor([t0], device=DEV)).cpu().numpy()
    assert relerr(y, g[tag + "/t%d" % t0]) < 1e-4
    if tag == "nf32d2_2x24x20":  # training-style [B] timesteps
        y = m(x, c, torch.tensor([5, 60])).cpu().numpy()
        assert relerr(y, g[tag + "/tvec"]) < 1e-4


def test_unet_layers_vs_oracle():
    """Per-layer parity against the float64 oracle (every tap point of the network)."""
    nf, depth, B, H, W = 32, 2, 2, 24, 20
    m, params = make_model(nf, depth, flags=_lib.FLAG_KEEP_ACTIVATIONS)
    lq, xT = O.synth_inputs(1234, B, H, W)
    taps = {}
    ref = O.unet_forward(params, xT, lq, 7, depth=depth, dtype=np.float64, taps=taps)
    y = m(torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV), 7).cpu().numpy()
    worst = {}
    for name, want in taps.items():
        if name == "time_emb":
            continue
        got = m.debug_tap(name).numpy()
        assert got.shape == want.shape, name
        worst[name] = relerr(got, want)
    bad = {k: v for k, v in worst.items() if not v < 5e-5}
    assert not bad, bad
    assert relerr(y, ref) < 5e-5


def test_winograd_vs_direct_network():
    """The production plan (Winograd F(4x4,3x3) from 128 channels, F(2x2,3x3) from 256) and the all-direct plan
    compute the same network; so does the F(2x2)-only plan."""
    B, H, W = 4, 128, 128
    m, _ = model(64, 4)
    md, _ = make_model(64, 4, flags=_lib.FLAG_NO_WINOGRAD)
    lq, xT = O.synth_inputs(7, B, H, W)
    x, c = torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV)
    a = m(x, c, 33).cpu().numpy()
    b = md(x, c, 33).cpu().numpy()
    assert relerr(a, b) < 2e-5
    m2, _ = make_model(64, 4, flags=_lib.FLAG_NO_WINOGRAD_F43)
    assert relerr(m2(x, c, 33).cpu().numpy(), b) < 2e-5
    del m2
    import ctypes as _c
    buf = _c.create_string_buffer(1 << 18)
    _lib.check(_lib.lib().irsde_plan_describe(m.engine().h, B, H, W, buf, len(buf)))
    assert b"winograd F4" in buf.value  # the production plan really takes the Winograd path at this size
    # a Rain100H-like odd size (reflect-padded to 176x160): the deepest level is 22x20 -> F(4x4) does not tile it and the
    # plan falls back to F(2x2,3x3) there
    B, H, W = 8, 170, 150
    lq, xT = O.synth_inputs(8, B, H, W)
    x, c = torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV)
    assert relerr(m(x, c, 61).cpu().numpy(), md(x, c, 61).cpu().numpy()) < 2e-5
    _lib.check(_lib.lib().irsde_plan_describe(m.engine().h, B, H, W, buf, len(buf)))
    assert b"winograd F2" in buf.value and b"winograd F4" in buf.value
    del md


def test_unet_mfma_vs_naive_full_resolution():
    """Size-independent cross-check at a BASELINE-sized input (B=4, 256x256, nf=64): the MFMA
    implicit-GEMM path and the VALU direct-convolution path compute the same network."""
    B, H, W = 4, 256, 256
    m, _ = model(64, 4)
    mn, _ = make_model(64, 4, flags=_lib.FLAG_NAIVE_CONV)
    lq, xT = O.synth_inputs(99, B, H, W)
    x, c = torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV)
    a = m(x, c, 42).cpu().numpy()
    b = mn(x, c, 42).cpu().numpy()
    assert np.isfinite(a).all()
    assert relerr(a, b) < 5e-5
    # batch independence (no cross-batch op): image 2 alone == image 2 in the batch
    a2 = m(x[2:3], c[2:3], 42).cpu().numpy()
    assert relerr(a2, a[2:3]) < 5e-5
    del mn


# ---------------------------------------------------------------------------------------------
# sampler level
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["s10_T100", "s50_T200"])
def test_reverse_step_kernel_vs_reference_golden(golden, tag):
    g = golden.steps
    ms, T = (10, 100) if tag == "s10_T100" else (50, 200)
    sde = P.IRSDE(ms, T, "cosine", 0.005, device=DEV)
    x, mu, eh = (torch.from_numpy(g[tag + "/" + k]).to(DEV) for k in ("x", "mu", "eps_hat"))
    z = torch.from_numpy(O.synth_noise(5, T, tuple(x.shape))).to(DEV)
    L = _lib.lib()
    B, C, H, W = x.shape
    for t in (1, 2, T // 2, T):
        for mode, key in (("sde", "sde"), ("ode", "ode"), ("posterior", "post")):
            xx = x.clone()
            row = sde._coef[t].contiguous()
            _lib.check(L.irsde_sde_step(_lib.MODE[mode], t, ctypes.c_void_p(row.data_ptr()),
                                        ctypes.c_void_p(xx.data_ptr()), ctypes.c_void_p(mu.data_ptr()),
                                        ctypes.c_void_p(eh.data_ptr()), ctypes.c_void_p(z[t].data_ptr()), 0, 0,
                                        B, C, H, W, None))
            torch.cuda.synchronize()
            np.testing.assert_allclose(xx.cpu().numpy(), g[tag + "/%s_t%d" % (key, t)], rtol=1e-5, atol=2e-6)


def _sample(m, mode, T, lq, xT, z, graph, sde=None):
    sde = sde or P.IRSDE(10, T, "cosine", 0.005, device=DEV)
    sde.set_model(m)
    sde.use_graph = graph
    sde.injected_noise = torch.from_numpy(z).to(DEV) if z is not None else None
    sde.set_mu(torch.from_numpy(lq).to(DEV))
    x_in = torch.from_numpy(xT).to(DEV)
    keep = x_in.clone()
    fn = {"sde": sde.reverse_sde, "ode": sde.reverse_ode, "posterior": sde.reverse_posterior}[mode]
    out = fn(x_in)
    assert torch.equal(x_in, keep)  # the input state is not modified (reference clones it)
    return out.cpu().numpy()


@pytest.mark.parametrize("tag,modes", [("nf32d2_2x16x16_T20", ["sde", "ode", "posterior"]),
                                       ("nf64d4_1x32x32_T100", ["sde", "ode", "posterior"]),
                                       ("nf64d4_1x128x128_T100", ["sde", "posterior"])])
def test_sampler_vs_reference_golden(golden, tag, modes):
    """End-to-end reverse samplers with injected noise vs the REAL reference's outputs
    (last case = BASELINE.json configs[0]: 1x3x128x128, T=100)."""
    g = golden.sampler
    nf, depth, B, H, W, T = (int(v) for v in g[tag + "/cfg"])
    m, _ = model(nf, depth)
    lq, xT = O.synth_inputs(1234, B, H, W)
    z = O.synth_noise(7, T, (B, 3, H, W))
    for mode in modes:
        ref = g[tag + "/" + mode]
        eager = _sample(m, mode, T, lq, xT, z, graph=False)
        assert np.isfinite(eager).all()
        assert relerr(eager, ref) < 2e-3, (tag, mode)
        # north_star: 1e-3 max-abs in fp32 on fixed noise seeds
        assert float(np.abs(eager.astype(np.float64) - ref).max()) < 1e-3, (tag, mode)
        graph = _sample(m, mode, T, lq, xT, z, graph=True)
        assert np.array_equal(eager, graph), "hipGraph replay must be bit-identical to eager launches"


def test_sampler_vs_oracle_small():
    """Same sampler vs the float64 oracle loop (tighter 'truth' than the fp32 reference)."""
    nf, depth, B, H, W, T = 32, 2, 2, 16, 16, 20
    m, params = model(nf, depth)
    lq, xT = O.synth_inputs(1234, B, H, W)
    z = O.synth_noise(7, T, (B, 3, H, W))
    sde = P.IRSDE(10, T, "cosine", 0.005, device=DEV)
    c = sde._coef.numpy()
    sch = dict(T=T, max_sigma=sde.max_sigma, dt=np.float32(float(sde.dt)), thetas=c[:, 0], sigmas=c[:, 1],
               sigma_bars=c[:, 2], x0_gain=c[:, 5], post_term1=c[:, 6], post_term2=c[:, 7], post_std=c[:, 8],
               thetas_cumsum=sde._cpu["thetas_cumsum"].numpy())
    for mode in ("sde", "ode", "posterior"):
        want = O.sample(params, sch, xT, lq, mode, noise=z, depth=depth, dtype=np.float64)
        got = _sample(m, mode, T, lq, xT, z, graph=True)
        assert relerr(got, want) < 1e-3, mode


def test_partial_T_and_segments():
    """reverse_*(xt, T=k) starts at step k (sde_utils.py:253-256); T..s then s..1 equals T..1."""
    nf, depth, B, H, W, T = 32, 2, 1, 16, 16, 20
    m, _ = model(nf, depth)
    lq, xT = O.synth_inputs(3, B, H, W)
    z = O.synth_noise(7, T, (B, 3, H, W))
    sde = P.IRSDE(10, T, "cosine", 0.005, device=DEV)
    full = _sample(m, "posterior", T, lq, xT, z, True, sde)
    L = _lib.lib()
    eng = m.engine()
    x = torch.from_numpy(xT).to(DEV)
    mu = torch.from_numpy(lq).to(DEV)
    zz = torch.from_numpy(z).to(DEV)
    mid = torch.empty_like(x)
    out = torch.empty_like(x)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.check(L.irsde_sample(eng.h, 2, p(x), p(mu), p(zz), 0, 0, B, H, W, T, 7, p(mid), None, 1))
    _lib.check(L.irsde_sample(eng.h, 2, p(mid), p(mu), p(zz), 0, 0, B, H, W, 7, 0, p(out), None, 1))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), full)


def test_schedule_change_invalidates_captured_graphs():
    """A graph captured under one schedule must not be replayed after IRSDE(T', ...) re-sets the tables."""
    nf, depth, B, H, W = 32, 2, 1, 16, 16
    m, _ = make_model(nf, depth)
    lq, xT = O.synth_inputs(5, B, H, W)
    outs = {}
    for T in (8, 5, 8):
        z = O.synth_noise(7, 8, (B, 3, H, W))
        outs.setdefault(T, []).append(_sample(m, "posterior", T, lq, xT, z, True))
    assert np.array_equal(outs[8][0], outs[8][1])
    eager = _sample(m, "posterior", 5, lq, xT, O.synth_noise(7, 8, (B, 3, H, W)), False)
    assert np.array_equal(outs[5][0], eager)


def test_foreign_model_path_matches_engine_path():
    """A score model that is NOT our ConditionalUNet is called per step like the reference does, with
    the fused HIP update kernel in between; with our UNet wrapped as an opaque callable both paths agree."""
    nf, depth, B, H, W, T = 32, 2, 2, 16, 16, 12
    m, _ = model(nf, depth)
    lq, xT = O.synth_inputs(5, B, H, W)
    z = O.synth_noise(7, T, (B, 3, H, W))
    a = _sample(m, "sde", T, lq, xT, z, True)

    class Opaque(torch.nn.Module):
        def forward(self, x, mu, t, **kw):
            return m(x, mu, t)
    b = _sample(Opaque(), "sde", T, lq, xT, z, True)
    assert relerr(b, a) < 1e-6


def test_philox_rng_vs_oracle_and_sharding_invariance():
    L = _lib.lib()
    B, CHW, t, seed = 3, 3 * 17 * 13, 9, 0x1234567890ABCDEF
    out = torch.empty(B, CHW, device=DEV)
    _lib.check(L.irsde_philox_normal(ctypes.c_void_p(out.data_ptr()), B, CHW, t, seed, 5, None))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for b in range(B):
        want = O.device_normal(seed, t, 5 + b, CHW)
        np.testing.assert_allclose(got[b], want, rtol=0, atol=2e-5)
    # image 6 drawn as "image 1 of a shard starting at 5" == "image 0 of a shard starting at 6"
    out2 = torch.empty(1, CHW, device=DEV)
    _lib.check(L.irsde_philox_normal(ctypes.c_void_p(out2.data_ptr()), 1, CHW, t, seed, 6, None))
    torch.cuda.synchronize()
    assert np.array_equal(out2.cpu().numpy()[0], got[1])
    assert abs(got.mean()) < 0.05 and abs(got.std() - 1) < 0.05


def test_sampler_philox_is_shard_invariant():
    """Without injected noise the sampler draws Philox noise keyed by the global image index:
    sampling images [0,4) at once == sampling [0,2) and [2,4) as two shards."""
    nf, depth, B, H, W, T = 32, 2, 4, 16, 16, 10
    m, _ = model(nf, depth)
    lq, xT = O.synth_inputs(11, B, H, W)
    sde = P.IRSDE(10, T, "cosine", 0.005, device=DEV)
    sde.seed = 77
    full = _sample(m, "sde", T, lq, xT, None, True, sde)
    parts = []
    for lo in (0, 2):
        sde.image_offset = lo
        parts.append(_sample(m, "sde", T, lq[lo:lo + 2], xT[lo:lo + 2], None, True, sde))
    sde.image_offset = 0
    assert relerr(np.concatenate(parts), full) < 1e-5
    other = P.IRSDE(10, T, "cosine", 0.005, device=DEV)
    other.seed = 78
    assert relerr(_sample(m, "sde", T, lq, xT, None, True, other), full) > 1e-3  # the seed matters


def test_denoising_model_boundary(tmp_path):
    """DenoisingModel.feed_data/test/get_current_visuals (deraining/models/denoising_model.py:121-171)
    driven exactly like deraining/test.py:104-112, loading a reference-format checkpoint."""
    nf, depth, H, W, T = 32, 2, 24, 20, 8
    params = O.synth_params(seed=0, nf=nf, depth=depth)
    ckpt = tmp_path / "G.pth"
    torch.save({"module." + k: torch.from_numpy(v) for k, v in params.items()}, ckpt)
    opt = {"model": "denoising", "is_train": False, "gpu_ids": [0], "dist": False,
           "network_G": {"which_model_G": "ConditionalUNet", "setting": {"in_nc": 3, "out_nc": 3, "nf": nf, "depth": depth}},
           "path": {"pretrain_model_G": str(ckpt), "strict_load": True}}
    mdl = P.create_model(opt)
    sde = P.IRSDE(max_sigma=10, T=T, schedule="cosine", eps=0.005, device=mdl.device)
    sde.set_model(mdl.model)
    lq, xT = O.synth_inputs(21, 1, H, W)
    z = O.synth_noise(7, T, (1, 3, H, W))
    sde.injected_noise = torch.from_numpy(z).to(mdl.device)
    mdl.feed_data(torch.from_numpy(xT), torch.from_numpy(lq), torch.from_numpy(lq))  # CPU tensors, like test.py
    sch = None
    for mode in ("posterior", "sde"):
        mdl.test(sde, mode=mode, save_states=False)
        vis = mdl.get_current_visuals()
        assert set(vis) == {"Input", "Output", "GT"} and vis["Output"].shape == (3, H, W)
        c = sde._coef.numpy()
        sch = dict(T=T, max_sigma=sde.max_sigma, dt=np.float32(float(sde.dt)), thetas=c[:, 0], sigmas=c[:, 1],
                   sigma_bars=c[:, 2], x0_gain=c[:, 5], post_term1=c[:, 6], post_term2=c[:, 7], post_std=c[:, 8])
        want = O.sample(params, sch, xT, lq, mode, noise=z, depth=depth, dtype=np.float64)
        assert relerr(vis["Output"].numpy()[None], want) < 1e-3
    # the deblurring / deshadow / inpainting / sisr copies of the wrapper: test(sde, save_states) == reverse_sde
    mdl2 = P.create_model(opt, task="deblurring")
    assert isinstance(mdl2, P.ReverseSDEDenoisingModel)
    sde.set_model(mdl2.model)
    mdl2.feed_data(torch.from_numpy(xT), torch.from_numpy(lq))
    mdl2.test(sde, save_states=False)
    assert relerr(mdl2.get_current_visuals(need_GT=False)["Output"].numpy(), vis["Output"].numpy()) < 1e-6  # vis: mode "sde"
    sde.set_model(mdl.model)
    # save_states dumps PNGs every T//100 (>=1) steps like sde_utils.py:260-264 and gives the same result
    out_a = sde.reverse_sde(mdl.state)
    out_b = sde.reverse_sde(mdl.state, save_states=True, save_dir=str(tmp_path / "sde_state"))
    assert torch.equal(out_a, out_b)
    assert len(list((tmp_path / "sde_state").glob("state_*.png"))) == T


# ---------------------------------------------------------------------------------------------
# ConditionalNAFNet (Refusion) — SURVEY.md §8(f) N1
# ---------------------------------------------------------------------------------------------
NAF_CFGS = {"refusion": dict(width=64, enc_blk_nums=[1, 1, 1, 28], middle_blk_num=1, dec_blk_nums=[1, 1, 1, 1]),
            "w32_e12": dict(width=32, enc_blk_nums=[1, 2], middle_blk_num=1, dec_blk_nums=[1, 1])}
_NAF = {}


def naf_model(name, flags=0):
    if (name, flags) not in _NAF:
        cfg = NAF_CFGS[name]
        params = O.naf_synth_params(seed=0, img_channel=3, width=cfg["width"], middle_blk_num=cfg["middle_blk_num"],
                                    enc_blk_nums=tuple(cfg["enc_blk_nums"]), dec_blk_nums=tuple(cfg["dec_blk_nums"]))
        m = P.ConditionalNAFNet(img_channel=3, **cfg)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
        m.engine_flags = flags
        _NAF[name, flags] = (m.to(DEV).eval(), params)
    return _NAF[name, flags]


@pytest.mark.parametrize("tag", ["w32_e12_2x24x20", "refusion_1x64x64", "refusion_2x40x56"])
def test_nafnet_forward_vs_reference_golden(golden, tag):
    g = golden.nafnet
    m, _ = naf_model("refusion" if tag.startswith("refusion") else "w32_e12")
    B, H, W = (int(v) for v in g[tag + "/shape"])
    lq, xT = O.synth_inputs(1234, B, H, W, max_sigma=50)
    x, c = torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV)
    for t in g[tag + "/ts"]:
        y = m(x, c, int(t)).cpu().numpy()
        assert relerr(y, g[tag + "/t%d" % t]) < 1e-4, (tag, int(t))


def test_nafnet_layers_vs_oracle():
    cfg = NAF_CFGS["w32_e12"]
    m, params = naf_model("w32_e12", flags=_lib.FLAG_KEEP_ACTIVATIONS)
    B, H, W = 2, 24, 20
    lq, xT = O.synth_inputs(1234, B, H, W, max_sigma=50)
    taps = {}
    ref = O.nafnet_forward(params, xT, lq, 7, tuple(cfg["enc_blk_nums"]), cfg["middle_blk_num"], tuple(cfg["dec_blk_nums"]),
                           dtype=np.float64, taps=taps)
    y = m(torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV), 7).cpu().numpy()
    bad = {}
    for name, want in taps.items():
        got = m.debug_tap(name).numpy()
        assert got.shape == want.shape, name
        e = relerr(got, want)
        if not e < 5e-5:
            bad[name] = e
    assert not bad, bad
    assert relerr(y, ref) < 5e-5
    # training-style per-sample timesteps
    y2 = m(torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV), torch.tensor([5, 60])).cpu().numpy()
    r2 = O.nafnet_forward(params, xT, lq, np.array([5, 60]), tuple(cfg["enc_blk_nums"]), cfg["middle_blk_num"],
                          tuple(cfg["dec_blk_nums"]), dtype=np.float64)
    assert relerr(y2, r2) < 5e-5


@pytest.mark.parametrize("tag,key,T", [("w32_e12_2x24x20", "sampler_2x24x20_T20", 20), ("refusion_1x64x64", "sampler_1x32x32_T100", 100)])
def test_nafnet_sampler_vs_reference_golden(golden, tag, key, T):
    """Refusion sampler (refusion.yml: max_sigma 50, cosine, eps 0.005) with injected noise vs the real reference."""
    g = golden.nafnet
    m, _ = naf_model("refusion" if tag.startswith("refusion") else "w32_e12")
    ref_sde = g["%s/%s/sde" % (tag, key)]
    B, _, H, W = ref_sde.shape
    lq, xT = O.synth_inputs(1234, B, H, W, max_sigma=50)
    z = O.synth_noise(7, T, (B, 3, H, W))
    sde = P.IRSDE(50, T, "cosine", 0.005, device=DEV)
    for mode in ("sde", "posterior"):
        got = _sample(m, mode, T, lq, xT, z, graph=True, sde=sde)
        assert relerr(got, g["%s/%s/%s" % (tag, key, mode)]) < 2e-3, (tag, mode)
        assert np.array_equal(got, _sample(m, mode, T, lq, xT, z, graph=False, sde=sde))


# ---------------------------------------------------------------------------------------------
# denoising-sde variant: DenoisingSDE + unconditional UNet with full attention — SURVEY.md §8(f) N2
# ---------------------------------------------------------------------------------------------
_DSDE = {}


def dsde_model(nf, depth, flags=0):
    if (nf, depth, flags) not in _DSDE:
        params = O.uncond_synth_params(seed=0, nf=nf, depth=depth)
        m = P.denoising_sde.ConditionalUNet(3, 3, nf, depth=depth)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
        m.engine_flags = flags
        _DSDE[nf, depth, flags] = (m.to(DEV).eval(), params)
    return _DSDE[nf, depth, flags]


@pytest.mark.parametrize("tag", ["nf32d2_2x24x20", "nf64d4_1x64x64", "nf64d4_2x88x80"])
def test_dsde_unet_forward_vs_reference_golden(golden, tag):
    """forward(x, time) of the denoising-sde UNet; nf64d4_2x88x80 has 120 bottleneck tokens (key/query tile masking)."""
    g = golden.dsde
    nf, depth, B, H, W = (int(v) for v in g[tag + "/cfg"])
    m, params = dsde_model(nf, depth)
    _, xT = O.synth_inputs(1234, B, H, W, max_sigma=25)
    x = torch.from_numpy(xT).to(DEV)
    for t in g[tag + "/ts"]:
        y = m(x, int(t)).cpu().numpy()
        assert relerr(y, g[tag + "/t%d" % t]) < 1e-4, (tag, int(t))


def test_dsde_full_attention_block_vs_oracle():
    nf, depth, B, H, W = 32, 2, 2, 24, 20
    m, params = dsde_model(nf, depth, flags=_lib.FLAG_KEEP_ACTIVATIONS)
    _, xT = O.synth_inputs(1234, B, H, W, max_sigma=25)
    taps = {}
    ref = O.uncond_unet_forward(params, xT, 7, depth=depth, taps=taps)
    y = m(torch.from_numpy(xT).to(DEV), 7).cpu().numpy()
    for name in ("mid_block1", "mid_attn"):
        assert relerr(m.debug_tap(name).numpy(), taps[name]) < 5e-5, name
    assert relerr(y, ref) < 5e-5


@pytest.mark.parametrize("tag,key", [("nf32d2_2x24x20", "sampler_2x16x16"), ("nf64d4_1x64x64", "sampler_1x32x32")])
def test_dsde_sampler_vs_reference_golden(golden, tag, key):
    """DenoisingSDE.reverse_ode / reverse_sde from the optimal timestep (denoising-sde test.py:103-107) vs the reference."""
    g = golden.dsde
    nf, depth = (int(v) for v in g[tag + "/cfg"][:2])
    m, _ = dsde_model(nf, depth)
    k = "%s/%s" % (tag, key)
    noisy = g[k + "/noisy"]
    sde = P.DenoisingSDE(max_sigma=75, T=100, device=DEV)
    # (tables are rebuilt with torch CPU ops on THIS host: equal to the reference's up to libm ulps amplified by the
    #  1 - exp(-x) cancellation at small t; bit-equality on the build host is pinned in tests/test_host_logic.py)
    np.testing.assert_allclose(float(sde.dt), float(g["sde/dt"]), rtol=1e-6)
    np.testing.assert_allclose(sde.sigma_bars.cpu().numpy()[1:], g["sde/sigma_bars"][1:], rtol=5e-4)
    sde.set_model(m)
    Topt = sde.get_optimal_timestep(25)
    assert int(Topt) == int(g[k + "/T"])
    sde.injected_noise = torch.from_numpy(O.synth_noise(7, 100, noisy.shape)).to(DEV)
    x = torch.from_numpy(noisy).to(DEV)
    for mode, fn in (("ode", sde.reverse_ode), ("sde", sde.reverse_sde)):
        got = fn(x, T=Topt).cpu().numpy()
        assert relerr(got, g[k + "/" + mode]) < 2e-3, (tag, mode)
        sde.use_graph = False
        assert np.array_equal(fn(x, T=Topt).cpu().numpy(), got)
        sde.use_graph = True
    # the x0-guided variant (real score) runs the per-step fused update
    x0 = torch.from_numpy(noisy * 0.5).to(DEV)
    assert torch.isfinite(sde.reverse_ode(x, x0=x0, T=5)).all()


# ---------------------------------------------------------------------------------------------
# evaluation tail
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["rgb_1x3x40x52_cb0", "rgb_2x3x64x48_cb4", "gray_1x1x33x37_cb0"])
def test_eval_metrics_vs_reference_golden(golden, tag):
    """Device tensor2img / PSNR / SSIM / Y-channel metrics vs the reference's functions: the uint8 image bit-exact,
    PSNR to 1e-12 (its squared-error sum is an exact integer), SSIM / Y metrics to 1e-10 (float64 summation order)."""
    g = golden.metrics
    B, C, H, W, cb = (int(v) for v in g[tag + "/cfg"])
    out, gt, want = torch.from_numpy(g[tag + "/out"]).to(DEV), torch.from_numpy(g[tag + "/gt"]).to(DEV), g[tag + "/metrics"]
    assert np.array_equal(P.metrics.tensor2img(out[0]), g[tag + "/img0"])
    m = P.metrics.evaluate_batch(out, gt, crop_border=cb)
    np.testing.assert_allclose(m["psnr"], want[:, 0], rtol=1e-12)
    np.testing.assert_allclose(m["ssim"], want[:, 1], rtol=1e-10)
    if C == 3:
        np.testing.assert_allclose(m["psnr_y"], want[:, 2], rtol=1e-10)
        np.testing.assert_allclose(m["ssim_y"], want[:, 3], rtol=1e-10)
    else:
        assert np.isnan(m["psnr_y"]).all() and np.isnan(m["ssim_y"]).all()
    avg = P.metrics.reduce_metrics({"psnr": m["psnr"], "ssim": m["ssim"]})
    assert abs(avg["psnr"] - want[:, 0].mean()) < 1e-9


def test_eval_metrics_full_size_properties():
    """BASELINE-sized batch (16 x 3 x 256 x 256): identical images give PSNR = inf and SSIM = 1; the metrics of image b
    do not depend on the rest of the batch; a sampled oracle check on one image."""
    rs = np.random.RandomState(3)
    gt = rs.rand(16, 3, 256, 256).astype(np.float32)
    out = np.clip(gt + 0.02 * rs.standard_normal(gt.shape), -0.1, 1.1).astype(np.float32)
    dg, do = torch.from_numpy(gt).to(DEV), torch.from_numpy(out).to(DEV)
    same = P.metrics.evaluate_batch(dg, dg)
    assert np.isinf(same["psnr"]).all() and np.allclose(same["ssim"], 1.0, atol=1e-12) and np.allclose(same["ssim_y"], 1.0, atol=1e-12)
    m = P.metrics.evaluate_batch(do, dg, crop_border=2)
    one = P.metrics.evaluate_batch(do[5:6], dg[5:6], crop_border=2)
    for k in m:
        assert m[k][5] == one[k][0]
    want = O.eval_tail(out[5], gt[5], 2)
    np.testing.assert_allclose([m["psnr"][5], m["ssim"][5], m["psnr_y"][5], m["ssim_y"][5]], want, rtol=1e-10)
    # gather path: BGR uint8 images of the whole batch in one call
    imgs = P.metrics.tensor2img_batch(do)
    assert imgs.shape == (16, 256, 256, 3) and np.array_equal(imgs[5], O.tensor2img(out[5]))


# ---------------------------------------------------------------------------------------------
# latent wrapper
# ---------------------------------------------------------------------------------------------
def latent_unet(cfg):
    m = P.latent.UNet(in_ch=3, out_ch=3, ch=cfg["ch"], ch_mult=list(cfg["ch_mult"]), embed_dim=cfg["embed_dim"])
    params = O.latent_unet_synth_params(seed=0, in_ch=3, out_ch=3, **cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    return m.to(DEV).eval(), params


@pytest.mark.parametrize("tag,cfg", [("nasde_1x3x40x52", dict(ch=8, ch_mult=(4, 8, 8, 16), embed_dim=8)),
                                     ("bokeh_2x3x24x32", dict(ch=16, ch_mult=(1, 2, 4), embed_dim=4))])
def test_latent_unet_vs_reference_golden(golden, tag, cfg):
    """UNet.encode / decode (latent-dehazing UNet_arch.py:59-91) vs the reference: latent, all 2*depth+1 skips, the
    reconstruction, and a decode from reference-made (perturbed latent, skips).  1e-4 of max|ref| like one network pass."""
    g = golden.latent
    B, H, W = (int(v) for v in g[tag + "/shape"])
    m, _ = latent_unet(cfg)
    lq, _ = O.synth_inputs(1234, B, H, W)
    lat, hid = m.encode(torch.from_numpy(lq).to(DEV))
    assert relerr(lat.cpu().numpy(), g[tag + "/latent"]) < 1e-4
    assert len(hid) == 2 * len(cfg["ch_mult"]) + 1
    for i, h in enumerate(hid):
        want = g[tag + "/hidden%d" % i]
        assert tuple(h.shape) == want.shape and relerr(h.cpu().numpy(), want) < 1e-4, i
    assert relerr(m.decode(lat, hid).cpu().numpy(), g[tag + "/decode"]) < 1e-4
    assert relerr(m(torch.from_numpy(lq).to(DEV)).cpu().numpy(), g[tag + "/decode"]) < 1e-4   # forward = decode(encode)
    hid_ref = [torch.from_numpy(g[tag + "/hidden%d" % i]).to(DEV) for i in range(len(hid))]
    rec2 = m.decode(torch.from_numpy(g[tag + "/latent2"]).to(DEV), hid_ref)
    assert tuple(rec2.shape) == (B, 3, H, W) and relerr(rec2.cpu().numpy(), g[tag + "/decode2"]) < 1e-4


def test_latent_hidden_stays_resident_and_never_dangles():
    """ABI 104: UNet.encode leaves the skips in the engine and returns a lazy list; decode(latent, that list) reads them in place.  Same bits as
    the NCHW round trip; touching the list, a second encode, or a decode with other skips materialises it first (the reference semantics: `hidden`
    is a plain list of tensors, UNet_arch.py:59-91)."""
    cfg = dict(ch=16, ch_mult=(1, 2, 4), embed_dim=4)
    m, _ = latent_unet(cfg)
    a = torch.from_numpy(O.synth_inputs(11, 2, 24, 32)[0]).to(DEV)
    b = torch.from_numpy(O.synth_inputs(12, 2, 24, 32)[0]).to(DEV)
    m.resident_hidden = False
    lat_a, hid_a = m.encode(a)
    rec_a = m.decode(lat_a, hid_a)
    lat_b, hid_b = m.encode(b)
    rec_b = m.decode(lat_b, hid_b)
    m.resident_hidden = True
    l1, h1 = m.encode(a)
    assert "resident" in repr(h1) and len(h1) == len(hid_a) and torch.equal(l1, lat_a)
    assert torch.equal(m.decode(l1, h1), rec_a)                     # in place: no NCHW tensors were made
    assert "resident" in repr(h1)
    l2, h2 = m.encode(b)                                            # overwrites the engine's skips: h1 gets its tensors first
    assert "NCHW" in repr(h1) and "resident" in repr(h2)
    assert all(torch.equal(x, y) for x, y in zip(h1, hid_a))
    assert torch.equal(m.decode(l1, h1), rec_a)                     # decode with OTHER skips: h2 is materialised before they are overwritten
    assert "NCHW" in repr(h2) and all(torch.equal(x, y) for x, y in zip(h2, hid_b))
    assert torch.equal(m.decode(l2, h2), rec_b)
    l3, h3 = m.encode(a)
    assert torch.equal(h3[1], hid_a[1]) and "NCHW" in repr(h3)      # indexing materialises
    l4, h4 = m.encode(b)
    as_list = list(h4)                                              # so does list(...) (a Sequence, not a list subclass)
    assert len(as_list) == len(hid_b) and all(torch.equal(x, y) for x, y in zip(as_list, hid_b))
    assert torch.equal(m.decode(l4, as_list), rec_b)
    # beyond Sequence the object turns into the reference's plain list, and a materialised one no longer pins the engine
    import io
    l5, h5 = m.encode(a)
    assert h5._eng is not None
    buf = io.BytesIO()
    torch.save(h5, buf)
    buf.seek(0)
    back = torch.load(buf, weights_only=False)
    assert type(back) is list and all(torch.equal(x, y) for x, y in zip(back, hid_a))
    assert h5._eng is None and "NCHW" in repr(h5)
    plus = h5 + [l5]
    assert type(plus) is list and len(plus) == len(hid_a) + 1 and type(h5.tolist()) is list
    l3, h3 = m.encode(a)
    # the C ABI refuses a decode without skips when none are resident (the decode above put the caller's skips in place)
    out = torch.empty_like(rec_a)
    rc = P._lib.lib().irsde_latent_decode(m.engine(a.device).h, ctypes.c_void_p(l3.data_ptr()), None, 2, 24, 32, ctypes.c_void_p(out.data_ptr()), P._lib.stream_ptr())
    assert rc == 0   # (encode(a) just made them resident again)
    assert torch.equal(out, rec_a)
    rc = P._lib.lib().irsde_latent_decode(m.engine(a.device).h, ctypes.c_void_p(l3.data_ptr()), None, 3, 24, 32, ctypes.c_void_p(out.data_ptr()), P._lib.stream_ptr())
    assert rc != 0 and b"preceding irsde_latent_encode" in P._lib.lib().irsde_last_error()


def test_latent_pipeline_vs_reference_golden(golden):
    """latent-dehazing/test.py:90-100 end to end through LatentDenoisingModel: encode -> noise_state -> reverse_sde / ode
    of the latent-task ConditionalNAFNet (ending(x + intro), 8 latent channels) -> decode, injected noise."""
    g = golden.latent
    opt = {"network_G": {"which_model": "ConditionalNAFNet",
                         "setting": dict(img_channel=8, width=32, enc_blk_nums=[1, 2], middle_blk_num=1, dec_blk_nums=[1, 1])},
           "network_L": {"which_model": "UNet", "setting": dict(in_ch=3, out_ch=3, ch=8, ch_mult=[4, 8, 8, 16], embed_dim=8)},
           "path": {}}
    model = P.LatentDenoisingModel(opt)
    nparams = O.naf_synth_params(seed=0, img_channel=8, width=32, middle_blk_num=1, enc_blk_nums=(1, 2), dec_blk_nums=(1, 1))
    model.model.load_state_dict({k: torch.from_numpy(v) for k, v in nparams.items()}, strict=True)
    uparams = O.latent_unet_synth_params(seed=0, in_ch=3, out_ch=3, ch=8, ch_mult=(4, 8, 8, 16), embed_dim=8)
    model.latent_model.load_state_dict({k: torch.from_numpy(v) for k, v in uparams.items()}, strict=True)
    # the score network alone (8-channel state, intro skip)
    xt, cond = torch.from_numpy(g["naf/xt"]).to(DEV), torch.from_numpy(g["naf/cond"]).to(DEV)
    for t in (4, 61):
        assert relerr(model.model(xt, cond, t).cpu().numpy(), g["naf/t%d" % t]) < 1e-4
    T = int(g["pipe/T"])
    sde = P.IRSDE(max_sigma=50, T=T, schedule="cosine", eps=0.005, device=DEV)
    sde.set_model(model.model)
    lq, _ = O.synth_inputs(1234, 1, 40, 52)
    latent_LQ, hidden = model.encode(torch.from_numpy(lq).to(DEV))
    noisy = latent_LQ + torch.from_numpy(g["pipe/z0"]).to(DEV) * sde.max_sigma
    sde.injected_noise = torch.from_numpy(O.synth_noise(7, T, tuple(latent_LQ.shape))).to(DEV)
    for mode in ("sde", "ode"):
        model.feed_data(noisy, latent_LQ)
        model.test(sde, hidden, perform_ode=(mode == "ode"))
        out = model.get_current_visuals(need_GT=False)["Output"].numpy()[None]
        assert relerr(out, g["pipe/out_" + mode]) < 2e-3, mode



def test_naf_chain_vs_per_layer_path():
    """r04: in the fp16 mode a run of 512-channel NAFBlocks on an 8 x 8 feature map (the 28-block level of BASELINE configs[4]) is ONE launch
    (csrc/naf_chain.hip: one work-group per image, activations in registers + LDS).  Against the per-layer path of the same mode
    (IRSDE_FLAG_NO_NAF_CHAIN) and the fp32 engine, on the latent-bokeh network (per-image lens FiLM, [B] timesteps) and the plain one."""
    from image_restoration_sde_amd import _lib
    rs = np.random.RandomState(5)
    for lens in (True, False):
        cls = P.latent_bokeh.ConditionalNAFNet if lens else P.ConditionalNAFNet
        kw = dict(img_channel=4, width=64, enc_blk_nums=[1, 1, 1, 3], middle_blk_num=1, dec_blk_nums=[1, 1, 1, 1])
        bp = O.naf_synth_params(seed=11, img_channel=4, width=64, middle_blk_num=1, enc_blk_nums=(1, 1, 1, 3), dec_blk_nums=(1, 1, 1, 1), lens=lens)
        for k in bp:   # default-init beta / gamma are zero (the blocks would be identities): give every branch weight
            if k.endswith(".beta") or k.endswith(".gamma"):
                bp[k] = (0.5 * rs.standard_normal(bp[k].shape)).astype(np.float32)
        xt = torch.from_numpy(rs.standard_normal((3, 4, 64, 64)).astype(np.float32)).to(DEV)
        cond = torch.from_numpy(rs.standard_normal((3, 4, 64, 64)).astype(np.float32)).to(DEV)
        li = [torch.from_numpy(rs.uniform(0.1, 1.0, 3).astype(np.float32)) for _ in range(3)]
        outs = {}
        for tag, flags in (("fp32", 0), ("chain", _lib.FLAG_FP16), ("layers", _lib.FLAG_FP16 | _lib.FLAG_NO_NAF_CHAIN)):
            m = cls(**kw)
            m.load_state_dict({k: torch.from_numpy(v) for k, v in bp.items()}, strict=True)
            m.engine_flags = flags
            m = m.to(DEV).eval()
            t = torch.tensor([5, 60, 33])
            outs[tag] = (m(xt, cond, t, lens_info=li) if lens else m(xt, cond, t)).cpu().numpy()
            if tag == "chain":
                one = (m(xt[1:2], cond[1:2], 60, lens_info=[v[1:2] for v in li]) if lens else m(xt[1:2], cond[1:2], 60)).cpu().numpy()
                assert relerr(one, outs[tag][1:2]) < 1e-6   # an image's result does not depend on the batch around it
        e_cl, e_c32, e_l32 = relerr(outs["chain"], outs["layers"]), relerr(outs["chain"], outs["fp32"]), relerr(outs["layers"], outs["fp32"])
        print("naf chain (lens=%s): chain vs layers %.3g, chain vs fp32 %.3g, layers vs fp32 %.3g" % (lens, e_cl, e_c32, e_l32))
        assert np.isfinite(outs["chain"]).all()
        assert e_cl < 3e-3 and e_c32 < 3e-3 and e_c32 < 2.0 * e_l32 + 1e-4


@pytest.mark.parametrize("lens", [False, True])
@pytest.mark.parametrize("enc3", [1, 3])
def test_naf_chain_blocks_vs_oracle(lens, enc3):
    """VERDICT r04 weak #1a: naf_chain_kernel against the ORACLE (oracle.naf_block = NAFBlock.forward, DenoisingNAFNet_arch.py:56-83) rather than
    against the repo's own per-layer path.  The chain's input and output are read back as taps (encoder level 3: `downs.2` -> `encoders.3`, decoder
    level 0: `ups.0` -> `decoders.0`); the oracle runs the same blocks in float64 from the SAME input with the fp16 operand restatement the conv tests
    use (`O.f16_convs`: operands of every 1x1 conv rounded to fp16, wide accumulation).  enc3 = 1: every chain launch is a single block (per-block
    check); enc3 = 3: a three-block run.  Bar 1e-3 of max|ref| per run (the chain additionally passes the conv1 output, the depthwise taps and the SCA
    vector through fp16: include/irsde_hip.h, IRSDE_FLAG_NO_NAF_CHAIN)."""
    rs = np.random.RandomState(5 + enc3)
    cls = P.latent_bokeh.ConditionalNAFNet if lens else P.ConditionalNAFNet
    encs = (1, 1, 1, enc3)
    kw = dict(img_channel=4, width=64, enc_blk_nums=list(encs), middle_blk_num=1, dec_blk_nums=[1, 1, 1, 1])
    bp = O.naf_synth_params(seed=11, img_channel=4, width=64, middle_blk_num=1, enc_blk_nums=encs, dec_blk_nums=(1, 1, 1, 1), lens=lens)
    for k in bp:   # default-init beta / gamma are zero (the blocks would be identities): give every branch weight
        if k.endswith(".beta") or k.endswith(".gamma"):
            bp[k] = (0.5 * rs.standard_normal(bp[k].shape)).astype(np.float32)
    m = cls(**kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in bp.items()}, strict=True)
    m.engine_flags = _lib.FLAG_FP16 | _lib.FLAG_KEEP_ACTIVATIONS
    m = m.to(DEV).eval()
    B = 2
    xt = torch.from_numpy(rs.standard_normal((B, 4, 64, 64)).astype(np.float32)).to(DEV)
    cond = torch.from_numpy(rs.standard_normal((B, 4, 64, 64)).astype(np.float32)).to(DEV)
    tvec = np.array([5, 60])
    li = [rs.uniform(0.1, 1.0, B).astype(np.float32) for _ in range(3)]
    if lens:
        m.set_lens_info([torch.from_numpy(v) for v in li], B, torch.device(DEV))
    buf = ctypes.create_string_buffer(1 << 16)
    _lib.check(_lib.lib().irsde_plan_describe(m.engine(torch.device(DEV)).h, B, 64, 64, buf, len(buf)))
    assert buf.value.count(b"naf_chain(fp16)") == 2, buf.value[-600:]     # encoder level 3 and decoder level 0 run as chains
    # (the forward comes AFTER the describe call: debug_tap reads the most recently used plan, and the [B]-timestep forward has its own)
    if lens:
        m(xt, cond, torch.from_numpy(tvec), lens_info=[torch.from_numpy(v) for v in li])
    else:
        m(xt, cond, torch.from_numpy(tvec))
    p64 = {k: np.asarray(v, dtype=np.float64) for k, v in bp.items()}
    temb, cam = O.naf_embeddings(p64, tvec, li if lens else None, np.float64)
    for src, dst, pres in (("downs.2", "encoders.3", ["encoders.3.%d." % j for j in range(enc3)]), ("ups.0", "decoders.0", ["decoders.0.0."])):
        x = m.debug_tap(src).numpy().astype(np.float64)
        assert x.shape == (B, 512, 8, 8), (src, x.shape)
        with O.f16_convs():
            for pre in pres:
                x = O.naf_block(p64, pre, x, temb, cam)
        got = m.debug_tap(dst).numpy()
        e = relerr(got, x)
        print("naf_chain vs oracle (lens=%s, %s, %d block(s)): %.3g" % (lens, dst, len(pres), e))
        assert np.isfinite(got).all() and e < 1e-3, (dst, e)


@pytest.mark.parametrize("B", [1, 3])
def test_nafnet_levels_64_32_16_vs_oracle(B):
    """VERDICT r04 weak #1a(iv): `dwconv_gate_kernel`'s column run adapts to the row width (late r04: 64 ... 16-pixel-wide latent maps); the whole-network
    goldens covered that only indirectly.  Per-level taps of an fp32 NAFNet on a 64 x 64 input — dwconv + SimpleGate + pooled sums at W = 64, 32, 16
    and 8 — against the float64 oracle at the per-layer bar (5e-5)."""
    encs, decs = (1, 1, 1, 1), (1, 1, 1, 1)
    params = O.naf_synth_params(seed=4, img_channel=4, width=32, middle_blk_num=1, enc_blk_nums=encs, dec_blk_nums=decs)
    rs = np.random.RandomState(17)
    for k in params:
        if k.endswith(".beta") or k.endswith(".gamma"):
            params[k] = (0.5 * rs.standard_normal(params[k].shape)).astype(np.float32)
    m = P.ConditionalNAFNet(img_channel=4, width=32, enc_blk_nums=list(encs), middle_blk_num=1, dec_blk_nums=list(decs))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m.engine_flags = _lib.FLAG_KEEP_ACTIVATIONS
    m = m.to(DEV).eval()
    xt = rs.standard_normal((B, 4, 64, 64)).astype(np.float32)
    cond = rs.standard_normal((B, 4, 64, 64)).astype(np.float32)
    taps = {}
    ref = O.nafnet_forward(params, xt, cond, 9, encs, 1, decs, dtype=np.float64, taps=taps)
    y = m(torch.from_numpy(xt).to(DEV), torch.from_numpy(cond).to(DEV), 9).cpu().numpy()
    bad = {}
    for name, want in taps.items():
        got = m.debug_tap(name).numpy()
        assert got.shape == want.shape, name
        e = relerr(got, want)
        if not e < 5e-5:
            bad[name] = e
    assert not bad, bad
    assert {taps["encoders.%d" % i].shape[-1] for i in range(4)} == {64, 32, 16, 8}
    assert relerr(y, ref) < 5e-5


@pytest.mark.parametrize("nsub", [2, 4])
def test_nafnet_sampler_concurrent_subbatches(nsub):
    """ABI 105: irsde_sample splits a chain-bearing ConditionalNAFNet batch into concurrent sub-batches (fork / join inside the captured step graph;
    engine_api.hip: one_step_split).  Forced here at B = 4 (the heuristic starts at 32 images): the split run must reproduce the un-split one — per-image
    lens FiLM rows, injected noise and the Philox streams are all addressed by the call-level image index — up to the tilings the smaller plans choose;
    graph replay and eager launches of the split step are bit-identical."""
    L = _lib.lib()
    rs = np.random.RandomState(23)
    kw = dict(img_channel=4, width=64, enc_blk_nums=[1, 1, 1, 2], middle_blk_num=1, dec_blk_nums=[1, 1, 1, 1])
    bp = O.naf_synth_params(seed=12, img_channel=4, width=64, middle_blk_num=1, enc_blk_nums=(1, 1, 1, 2), dec_blk_nums=(1, 1, 1, 1), lens=True)
    for k in bp:
        if k.endswith(".beta") or k.endswith(".gamma"):
            bp[k] = (0.3 * rs.standard_normal(bp[k].shape)).astype(np.float32)
    m = P.latent_bokeh.ConditionalNAFNet(**kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in bp.items()}, strict=True)
    m.set_compute_dtype("fp16")
    m = m.to(DEV).eval()
    B, T = 4, 5
    xt = torch.from_numpy(rs.standard_normal((B, 4, 64, 64)).astype(np.float32)).to(DEV)
    cond = torch.from_numpy(rs.standard_normal((B, 4, 64, 64)).astype(np.float32)).to(DEV)
    li = [torch.from_numpy(rs.uniform(0.1, 1.0, B).astype(np.float32)) for _ in range(3)]
    sde = P.IRSDE(max_sigma=50, T=T, schedule="cosine", eps=0.005, device=DEV)
    sde.set_model(m)
    sde.set_mu(cond)
    sde.image_offset = 3
    z = torch.from_numpy(O.synth_noise(9, T, (B, 4, 64, 64))).to(DEV)
    outs = {}
    try:
        for n in (1, nsub):
            L.irsde_debug_force_subbatches(n)
            for tag, noise, seed in (("inj", z, 0), ("rng", None, 5)):
                sde.injected_noise, sde.seed = noise, seed
                for graph in (True, False):
                    sde.use_graph = graph
                    outs[n, tag, graph] = sde.reverse_sde(xt, lens_info=li).cpu().numpy()
    finally:
        L.irsde_debug_force_subbatches(0)
        sde.use_graph = True
    for tag in ("inj", "rng"):
        assert np.isfinite(outs[nsub, tag, True]).all()
        assert np.array_equal(outs[nsub, tag, True], outs[nsub, tag, False]), tag        # captured fork / join == eager cross-stream dependencies
        e = relerr(outs[nsub, tag, True], outs[1, tag, True])
        print("NAFNet sampler, %d concurrent sub-batches vs one batch (%s): %.3g" % (nsub, tag, e))
        assert e < 2e-4, (tag, e)   # fp16-operand plans of B / nsub and B images: same roundings, different f32 summation orders through T steps
    # the images really differ from each other (a wrong sub-batch offset into the lens / noise tables would go unnoticed otherwise)
    assert relerr(outs[1, "rng", True][0], outs[1, "rng", True][B - 1]) > 1e-2


def test_latent_bokeh_nafnet_vs_reference_golden(golden):
    """latent-bokeh ConditionalNAFNet (lens_info kwargs, IRSDE_FLAG_NAF_LENS): forward with per-image lens triples and [B]
    timesteps, the reference's one-image int-time call, and reverse_sde(..., lens_info=...) on the engine fast path."""
    g = golden.latent
    m = P.latent_bokeh.ConditionalNAFNet(img_channel=4, width=32, enc_blk_nums=[1, 2], middle_blk_num=1, dec_blk_nums=[1, 1])
    bp = O.naf_synth_params(seed=3, img_channel=4, width=32, middle_blk_num=1, enc_blk_nums=(1, 2), dec_blk_nums=(1, 1), lens=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in bp.items()}, strict=True)
    m = m.to(DEV).eval()
    xt, cond, lens = torch.from_numpy(g["bokeh/xt"]).to(DEV), torch.from_numpy(g["bokeh/cond"]).to(DEV), g["bokeh/lens"]
    li = [torch.from_numpy(lens[:, i].copy()) for i in range(3)]
    y = m(xt, cond, torch.tensor([5, 60]), lens_info=li).cpu().numpy()
    assert relerr(y, g["bokeh/tvec"]) < 1e-4
    li1 = [torch.from_numpy(lens[:1, i].copy()) for i in range(3)]
    assert relerr(m(xt[:1], cond[:1], 33, lens_info=li1).cpu().numpy(), g["bokeh/t33"]) < 1e-4
    with pytest.raises(P.IrsdeError):
        m(xt, cond, 5)                                   # the reference raises KeyError without lens_info; loud here too
    T = int(g["bokeh/T"])
    sde = P.IRSDE(max_sigma=50, T=T, schedule="cosine", eps=0.005, device=DEV)
    sde.set_model(m)
    sde.set_mu(cond[:1])
    sde.injected_noise = torch.from_numpy(O.synth_noise(9, T, (1, 4, 12, 10))).to(DEV)
    out = sde.reverse_sde(xt[:1], lens_info=li1).cpu().numpy()
    assert relerr(out, g["bokeh/sde"]) < 2e-3
    # batch of two with different lens triples == the two single-image runs (per-image FiLM rows)
    sde.set_mu(cond)
    sde.injected_noise = None
    sde.seed = 3
    both = sde.reverse_ode(xt, lens_info=li).cpu().numpy()
    sde.set_mu(cond[1:2])
    sde.image_offset = 1
    one = sde.reverse_ode(xt[1:2], lens_info=[t[1:2] for t in li]).cpu().numpy()
    assert relerr(one, both[1:2]) < 1e-4


def test_group_sum_probe_toolchain_guard(tmp_path):
    """the cross-lane sums of the fp32 LayerNorm / attention / chain kernels (DPP + v_permlane16/32_swap inline assembly with a hand-placed
    s_nop) are opaque to LLVM's hazard recognizer.  tools/probe/group_sum_probe.hip holds the same helpers next to the __shfl_xor butterflies they replace:
    built with THIS toolchain and run on THIS GPU, every width must agree — a hipcc bump that breaks the wait-state assumption fails here by name
    (module_util.py:20-26 LayerNorm is the first caller)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "probe", "group_sum_probe.hip")
    exe = str(tmp_path / "group_sum_probe")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", src, "-o", exe], check=True, capture_output=True, timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


@pytest.mark.parametrize("B", [2, 8, 11])
def test_naf_chain_groups_per_image_equal_one_group(B):
    """r06 (ABI 107, csrc/naf_chain.hip): the NAFBlock chain (DenoisingNAFNet_arch.py:56-83) on 2 / 4 work-groups per image — each group owns a slice of the
    output channels, streams 1 / G of the weights and trades operand slices with the image's other groups through L2 (five barriers per block) — must give
    the ONE-group kernel's result bit for bit: same operations in the same order (the LayerNorms run on the whole gathered image in the one-group lane
    layout).  B = 2 / 8 / 11: one partly filled group of eight block-id residues, one full, two with a ragged second one; three-block and one-block runs;
    lens FiLM on.  The plan must say which kernel it launches."""
    rs = np.random.RandomState(40 + B)
    encs = (1, 1, 1, 3)
    kw = dict(img_channel=4, width=64, enc_blk_nums=list(encs), middle_blk_num=1, dec_blk_nums=[1, 1, 1, 1])
    bp = O.naf_synth_params(seed=12, img_channel=4, width=64, middle_blk_num=1, enc_blk_nums=encs, dec_blk_nums=(1, 1, 1, 1), lens=True)
    for k in bp:
        if k.endswith(".beta") or k.endswith(".gamma"):
            bp[k] = (0.5 * rs.standard_normal(bp[k].shape)).astype(np.float32)
    xt = torch.from_numpy(rs.standard_normal((B, 4, 64, 64)).astype(np.float32)).to(DEV)
    cond = torch.from_numpy(rs.standard_normal((B, 4, 64, 64)).astype(np.float32)).to(DEV)
    li = [torch.from_numpy(rs.uniform(0.1, 1.0, B).astype(np.float32)) for _ in range(3)]
    t = torch.from_numpy(rs.randint(1, 90, B))
    L = _lib.lib()
    outs = {}
    try:
        for g in (1, 2, 4):
            L.irsde_debug_force_chain_groups(g)
            m = P.latent_bokeh.ConditionalNAFNet(**kw)
            m.load_state_dict({k: torch.from_numpy(v) for k, v in bp.items()}, strict=True)
            m.engine_flags = _lib.FLAG_FP16
            m = m.to(DEV).eval()
            outs[g] = m(xt, cond, t, lens_info=li).cpu().numpy()
            buf = ctypes.create_string_buffer(1 << 16)
            _lib.check(L.irsde_plan_describe(m.engine(torch.device(DEV)).h, B, 64, 64, buf, len(buf)))
            assert buf.value.count(b"naf_chain(fp16)") == 2 and buf.value.count(b"groups=%d " % g) == 2, buf.value[-800:]   # encoder level 3 (three blocks), decoder level 0 (one)
    finally:
        L.irsde_debug_force_chain_groups(0)
    assert np.isfinite(outs[1]).all()
    for g in (2, 4):
        d = float(np.abs(outs[g] - outs[1]).max())
        print("naf chain B=%d: %d groups per image vs one: max diff %.3g (bit-identical: %s)" % (B, g, d, np.array_equal(outs[g], outs[1])))
        assert np.array_equal(outs[g], outs[1]), (B, g, d)


@pytest.mark.parametrize("B", [8, 5])
def test_naf_chain_groups_sampler_graph_replay(B):
    """r06: the split chain inside irsde_sample — ONE captured step graph replayed T times (the kernel restores its barrier counters itself: no memset node),
    two calls in a row (second call = pure replay), eager as well: 4 groups per image must reproduce the one-group sampler bit for bit, and no call may
    leave the co-residency error word set (sde_utils.py:252-266 is the loop)."""
    rs = np.random.RandomState(70 + B)
    kw = dict(img_channel=4, width=64, enc_blk_nums=[1, 1, 1, 2], middle_blk_num=1, dec_blk_nums=[1, 1, 1, 1])
    bp = O.naf_synth_params(seed=13, img_channel=4, width=64, middle_blk_num=1, enc_blk_nums=(1, 1, 1, 2), dec_blk_nums=(1, 1, 1, 1), lens=True)
    for k in bp:
        if k.endswith(".beta") or k.endswith(".gamma"):
            bp[k] = (0.5 * rs.standard_normal(bp[k].shape)).astype(np.float32)
    mu = torch.from_numpy(rs.rand(B, 4, 64, 64).astype(np.float32)).to(DEV)
    li = [torch.from_numpy(rs.uniform(0.1, 1.0, B).astype(np.float32)) for _ in range(3)]
    xT = mu + torch.from_numpy((0.5 * rs.standard_normal((B, 4, 64, 64))).astype(np.float32)).to(DEV)   # (one terminal state for both engines)
    L = _lib.lib()
    res = {}
    try:
        for g in (1, 4):
            L.irsde_debug_force_chain_groups(g)
            m = P.latent_bokeh.ConditionalNAFNet(**kw)
            m.load_state_dict({k: torch.from_numpy(v) for k, v in bp.items()}, strict=True)
            m.engine_flags = _lib.FLAG_FP16
            m = m.to(DEV).eval()
            sde = P.IRSDE(max_sigma=50, T=100, schedule="cosine", eps=0.005, device=torch.device(DEV))
            sde.set_model(m)
            sde.seed = 11
            sde.set_mu(mu)
            outs = []
            for graph in (True, True, False):
                sde.use_graph = graph
                outs.append(sde.reverse_sde(xT.clone(), T=6, lens_info=li).cpu().numpy())
            sde.reverse_sde(xT.clone(), T=1, lens_info=li)   # (a call after the last one: this is where a co-residency error of the previous call would be raised)
            torch.cuda.synchronize()
            assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2]), g   # replay = first launch = eager
            res[g] = outs[0]
    finally:
        L.irsde_debug_force_chain_groups(0)
    assert np.isfinite(res[1]).all()
    assert np.array_equal(res[4], res[1]), float(np.abs(res[4] - res[1]).max())


def test_naf_chain_spin_limit_ends_the_launch():
    """r06: the safety net of the split NAFBlock chain.  A group that never arrives (PROBES build, irsde_bench_naf_chain variant 26: group 1 of every image
    leaves at once — what a work-group that is not resident looks like to the others) must not hang the GPU: every waiting group gives up after its spin limit,
    the launch ENDS (seconds, not the harness's kill), the call reports the co-residency timeout, and the next, healthy launch runs and is correct in speed
    and result state (the counters were left dirty: the error path re-zeroes them)."""
    import time
    Lp = _lib.probes_lib()
    ms = ctypes.c_double()
    t0 = time.time()
    rc = Lp.irsde_bench_naf_chain(26, 2, 8, 1, ctypes.byref(ms))
    dt = time.time() - t0
    msg = Lp.irsde_last_error().decode()
    print("sabotaged launch: rc %d after %.1f s: %s" % (rc, dt, msg[:120]))
    assert rc != 0 and "co-resident" in msg, (rc, msg)
    assert dt < 60.0, dt
    rc = Lp.irsde_bench_naf_chain(24, 2, 8, 3, ctypes.byref(ms))
    assert rc == 0 and 0.0 < ms.value < 5.0, (rc, ms.value, Lp.irsde_last_error())


def test_naf_lnconv_kernel_vs_two_kernel_path(tmp_path):
    """r06: naf_lnconv_kernel (csrc/kernels_misc.hip: NAFBlock norm + FiLM + 1x1 convolution, SCA-scaled conv3, conv5 as one-piece-K launches in the fp16 operand mode,
    DenoisingNAFNet_arch.py:56-83) against the path it replaces (layernorm_kernel + conv_igemm_kernel: IRSDE_NAF_LNCONV=0, only reachable under IRSDE_TUNING=1, so the
    old path runs in a child process).  Same LayerNorm arithmetic, same fp16 operand rounding, same k order on the same MFMA, but not bit-identical: the
    epilogues are compiled differently (conv_igemm.hip with FMA contraction, kernels_misc.hip without), and a last-bit fp32 difference now and then flips the
    fp16 rounding of a later operand.  So the bar is relative to the fp32 engine on the same input: the forward of a lens-conditioned latent NAFNet (every level
    c = 64 / 128 / 256 outside the chain, [B] timesteps, B = 3 with ragged 64-pixel tiles at the deep levels) through the fused kernels is no further from the
    fp32 engine than through the two-kernel path (measured 4.25e-4 / 3.95e-4 of max|out|; 1.8e-4 between the two), and the two agree within 5e-4."""
    import subprocess
    import sys
    script = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
import image_restoration_sde_amd as P
from image_restoration_sde_amd import _lib
from oracle import irsde_oracle as O
rs = np.random.RandomState(91)
kw = dict(img_channel=4, width=64, enc_blk_nums=[1, 1, 1, 1], middle_blk_num=1, dec_blk_nums=[1, 1, 1, 1])
bp = O.naf_synth_params(seed=14, img_channel=4, width=64, middle_blk_num=1, enc_blk_nums=(1, 1, 1, 1), dec_blk_nums=(1, 1, 1, 1), lens=True)
for k in bp:
    if k.endswith(".beta") or k.endswith(".gamma"):
        bp[k] = (0.5 * rs.standard_normal(bp[k].shape)).astype(np.float32)
m = P.latent_bokeh.ConditionalNAFNet(**kw)
m.load_state_dict({k: torch.from_numpy(v) for k, v in bp.items()}, strict=True)
m.engine_flags = _lib.FLAG_FP16 if sys.argv[2] == "fp16" else 0
m = m.to("cuda:0").eval()
B = 3
xt = torch.from_numpy(rs.standard_normal((B, 4, 48, 40)).astype(np.float32)).cuda()
cond = torch.from_numpy(rs.standard_normal((B, 4, 48, 40)).astype(np.float32)).cuda()
li = [torch.from_numpy(rs.uniform(0.1, 1.0, B).astype(np.float32)) for _ in range(3)]
out = m(xt, cond, torch.tensor([3, 50, 97]), lens_info=li).cpu().numpy()
np.save(sys.argv[1], out)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tag, mode, env in (("fused", "fp16", {}), ("two_kernels", "fp16", {"IRSDE_TUNING": "1", "IRSDE_NAF_LNCONV": "0"}), ("fp32", "fp32", {})):
        f = str(tmp_path / (tag + ".npy"))
        r = subprocess.run([sys.executable, "-c", script, f, mode], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = np.load(f)
    e = relerr(res["fused"], res["two_kernels"])
    ef, eo = relerr(res["fused"], res["fp32"]), relerr(res["two_kernels"], res["fp32"])
    print("naf_lnconv_kernel vs layernorm_kernel + conv_igemm_kernel: rel err %.3g, bit-identical: %s; against the fp32 engine: %.3g (fused) / %.3g (two kernels)"
          % (e, np.array_equal(res["fused"], res["two_kernels"]), ef, eo))
    assert ef < 1.25 * eo + 1e-5, (ef, eo)     # the fused path is no further from the fp32 engine than the path it replaces
    assert np.isfinite(res["fused"]).all() and e < 5e-4, e
