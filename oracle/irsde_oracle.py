"""CPU oracle for the IR-SDE reverse sampler hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-numpy restatement of the reference algorithm
(Algolzw/image-restoration-sde).  It is the *checker* for the HIP path; it is
never imported by the product package (`image_restoration_sde_amd`).  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may use it.

Parity pin: every function here is checked against golden vectors produced by
the real reference code (imported from /root/reference on CPU) by
`oracle/gen_golden.py`; see `tests/test_oracle_golden.py`.

Each function cites the reference file:line it restates (paths relative to
/root/reference).  The arithmetic dtype is a parameter: float32 mirrors the
reference's dtype, float64 gives a "truth" both the reference (fp32, oneDNN
summation order) and the HIP path (fp32, MFMA summation order) are compared to.
"""
import math

import numpy as np

# --------------------------------------------------------------------------------------
# SDE schedule — codes/utils/sde_utils.py:84-152
# --------------------------------------------------------------------------------------


def irsde_schedule(max_sigma, T=100, schedule="cosine", eps=0.01):
    """Restates IRSDE.__init__/_initialize (sde_utils.py:84-152) in float32 numpy.

    Returns dict(max_sigma, dt, thetas, sigmas, thetas_cumsum, sigma_bars); arrays have
    length T+1 (index 0 is never used by the samplers, sde_utils.py:81-83).
    """
    f32 = np.float32
    max_sigma = max_sigma / 255 if max_sigma >= 1 else max_sigma  # :86
    if schedule == "cosine":  # :110-121
        timesteps = T + 2
        steps = timesteps + 1
        x = np.linspace(0, timesteps, steps, dtype=np.float64).astype(f32)
        s = 0.008
        inner = ((x / f32(timesteps)) + f32(s)) / f32(1 + s) * f32(math.pi) * f32(0.5)
        ac = np.cos(inner.astype(f32)).astype(f32) ** 2
        ac = (ac / ac[0]).astype(f32)
        thetas = (f32(1) - ac[1:-1]).astype(f32)
    elif schedule == "linear":  # :100-108
        timesteps = T + 1
        scale = 1000 / timesteps
        thetas = np.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=np.float64).astype(f32)
    elif schedule == "constant":  # :91-98
        thetas = np.ones(T + 1, dtype=f32)
    else:
        raise ValueError("unknown schedule %r" % (schedule,))
    sigmas = np.sqrt(f32(max_sigma ** 2 * 2) * thetas).astype(f32)  # :126-127
    thetas_cumsum = (np.cumsum(thetas, dtype=f32) - thetas[0]).astype(f32)  # :142
    dt = f32(f32(-1) / thetas_cumsum[-1]) * f32(math.log(eps))  # :143
    dt = f32(dt)
    sigma_bars = np.sqrt(
        f32(max_sigma ** 2) * (f32(1) - np.exp(f32(-2) * thetas_cumsum * dt).astype(f32))
    ).astype(f32)  # :129-130
    return dict(max_sigma=max_sigma, T=T, dt=dt, thetas=thetas, sigmas=sigmas,
                thetas_cumsum=thetas_cumsum, sigma_bars=sigma_bars)


def posterior_coeffs(sch, t):
    """(x0_gain, term1, term2, std) of reverse_posterior_step — sde_utils.py:197-223,237-239.

    term1/term2/std are ratios of fp32 cancellations ((1-C^2)/(1-B^2) with B,C -> 1 at small t):
    a 1-ulp difference between numpy's and torch's expf moves them by ~2e-4 relative, so when the
    caller supplies the tables the reference itself produced (keys post_term1/post_term2/post_std/
    x0_gain, e.g. from tests/golden/schedule.npz or from the product's torch-built tables, which
    are pinned bit-exactly to the golden ones) those are used verbatim."""
    f32 = np.float32
    if "post_term1" in sch:
        return (f32(sch["x0_gain"][t]), f32(sch["post_term1"][t]), f32(sch["post_term2"][t]),
                f32(sch["post_std"][t]))
    th, cs, dt = sch["thetas"], sch["thetas_cumsum"], sch["dt"]
    A = np.exp(-th[t] * dt, dtype=f32)
    B = np.exp(-cs[t] * dt, dtype=f32)
    C = np.exp(-cs[t - 1] * dt, dtype=f32)
    term1 = f32(A * (f32(1) - C ** 2) / (f32(1) - B ** 2))
    term2 = f32(C * (f32(1) - A ** 2) / (f32(1) - B ** 2))
    A2 = np.exp(f32(-2) * th[t] * dt, dtype=f32)
    B2 = np.exp(f32(-2) * cs[t] * dt, dtype=f32)
    C2 = np.exp(f32(-2) * cs[t - 1] * dt, dtype=f32)
    var = f32((f32(1) - A2) * (f32(1) - C2) / (f32(1) - B2))
    min_value = f32(f32(1e-20) * dt)
    logv = np.log(np.maximum(var, min_value), dtype=f32)
    std = f32(np.exp(f32(0.5) * logv, dtype=f32) * f32(sch["max_sigma"]))
    x0_gain = np.exp(cs[t] * dt, dtype=f32)
    return x0_gain, term1, term2, std


# ---- one reverse step, elementwise (float32 like the reference, or float64) ----------


def reverse_sde_step(sch, x, mu, noise, z, t, dtype=np.float32):
    """SDE.reverse_sde_step + IRSDE.sde_reverse_drift/dispersion/get_score_from_noise —
    sde_utils.py:44-45,175-176,181-182,184-185.  `z` is the injected N(0,1) draw that
    replaces torch.randn_like(x)."""
    d = dtype
    x, mu, noise, z = (np.asarray(a, dtype=d) for a in (x, mu, noise, z))
    theta, sigma, sbar, dt = d(sch["thetas"][t]), d(sch["sigmas"][t]), d(sch["sigma_bars"][t]), d(sch["dt"])
    score = -noise / sbar
    drift = (theta * (mu - x) - sigma ** 2 * score) * dt
    disp = sigma * (z * d(math.sqrt(float(sch["dt"]))))
    return (x - drift - disp).astype(d)


def reverse_ode_step(sch, x, mu, noise, t, dtype=np.float32):
    """SDE.reverse_ode_step + IRSDE.ode_reverse_drift — sde_utils.py:47-48,178-179."""
    d = dtype
    x, mu, noise = (np.asarray(a, dtype=d) for a in (x, mu, noise))
    theta, sigma, sbar, dt = d(sch["thetas"][t]), d(sch["sigmas"][t]), d(sch["sigma_bars"][t]), d(sch["dt"])
    score = -noise / sbar
    drift = (theta * (mu - x) - d(0.5) * sigma ** 2 * score) * dt
    return (x - drift).astype(d)


def reverse_posterior_step(sch, x, mu, noise, z, t, dtype=np.float32):
    """IRSDE.reverse_posterior_step — sde_utils.py:219-223 (+197-217, 237-239)."""
    d = dtype
    x, mu, noise, z = (np.asarray(a, dtype=d) for a in (x, mu, noise, z))
    g, t1, t2, std = (d(v) for v in posterior_coeffs(sch, t))
    sbar = d(sch["sigma_bars"][t])
    x0 = (x - mu - sbar * noise) * g + mu
    mean = t1 * (x - mu) + t2 * (x0 - mu) + mu
    return (mean + std * z).astype(d)


# --------------------------------------------------------------------------------------
# ConditionalUNet — codes/config/deraining/models/modules/{DenoisingUNet_arch,module_util}.py
# --------------------------------------------------------------------------------------


def round_bf16(a):
    """Round to the nearest bfloat16 (ties to even), returned in the dtype of `a`."""
    a = np.asarray(a)
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    u = (u + (((u >> 16) & 1) + np.uint32(0x7FFF))) & np.uint32(0xFFFF0000)
    return u.view(np.float32).astype(a.dtype)


def round_f16(a):
    """Round to the nearest IEEE binary16 (ties to even; the engine's IRSDE_FLAG_FP16 operand rounding), returned in the
    dtype of `a`."""
    a = np.asarray(a)
    return np.ascontiguousarray(a, dtype=np.float32).astype(np.float16).astype(a.dtype)


CONV_OPERANDS_F16 = False   # restatement of the engine's IRSDE_FLAG_FP16 mode: conv operands rounded to fp16, fp32+ accumulation
CONV_OPERANDS_BF16 = False  # restatement of the engine's IRSDE_FLAG_BF16 mode: conv operands rounded, fp32+ accumulation


ACT_STORAGE_BF16 = False    # restatement of IRSDE_FLAG_BF16_ACT: every stored activation tensor is rounded to bf16


class bf16_convs:
    """Context manager: every conv2d inside rounds activations and weights to bf16 first (the Linear layers of the
    time MLP, LayerNorm, attention and the update step stay in full precision, as in the engine).  store_bf16=True also
    rounds every tensor the engine keeps in HBM between kernels (IRSDE_FLAG_BF16_ACT): fused-epilogue conv outputs,
    LayerNorm outputs, the attention core's output."""

    def __init__(self, store_bf16=False):
        self.store = store_bf16

    def __enter__(self):
        global CONV_OPERANDS_BF16, ACT_STORAGE_BF16
        self.prev = (CONV_OPERANDS_BF16, ACT_STORAGE_BF16)
        CONV_OPERANDS_BF16, ACT_STORAGE_BF16 = True, self.store

    def __exit__(self, *a):
        global CONV_OPERANDS_BF16, ACT_STORAGE_BF16
        CONV_OPERANDS_BF16, ACT_STORAGE_BF16 = self.prev


class f16_convs:
    """Context manager: every conv2d inside rounds activations and weights to IEEE fp16 first (IRSDE_FLAG_FP16); everything
    else stays in full precision, as in the engine."""

    def __enter__(self):
        global CONV_OPERANDS_F16
        self.prev = CONV_OPERANDS_F16
        CONV_OPERANDS_F16 = True

    def __exit__(self, *a):
        global CONV_OPERANDS_F16
        CONV_OPERANDS_F16 = self.prev


def _st(x):
    """A tensor the engine writes to HBM (identity unless the bf16 storage mode is being restated)."""
    return round_bf16(x) if ACT_STORAGE_BF16 else x


def conv2d(x, w, b=None, stride=1, pad=0):
    """nn.Conv2d forward (cross-correlation), NCHW / OIHW, zero padding."""
    if CONV_OPERANDS_BF16:
        x, w = round_bf16(x), round_bf16(w)
    elif CONV_OPERANDS_F16:
        x, w = round_f16(x), round_f16(w)
    B, C, H, W = x.shape
    O, C2, kh, kw = w.shape
    assert C == C2
    if pad:
        x = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    win = np.lib.stride_tricks.sliding_window_view(x, (kh, kw), axis=(2, 3))
    win = win[:, :, ::stride, ::stride]  # B,C,Ho,Wo,kh,kw
    Ho, Wo = win.shape[2], win.shape[3]
    a = np.ascontiguousarray(win.transpose(0, 2, 3, 1, 4, 5)).reshape(B * Ho * Wo, C * kh * kw)
    out = a @ w.reshape(O, C * kh * kw).T
    out = out.reshape(B, Ho, Wo, O).transpose(0, 3, 1, 2)
    if b is not None:
        out = out + b.reshape(1, O, 1, 1)
    return np.ascontiguousarray(out)


def silu(x):
    """nn.SiLU — module_util.py:62-63."""
    return x / (1 + np.exp(-x))


def gelu(x):
    """nn.GELU (erf form) — DenoisingUNet_arch.py:45."""
    from math import erf
    verf = np.vectorize(erf, otypes=[np.float64])
    return (0.5 * x * (1 + verf(x.astype(np.float64) / math.sqrt(2.0)))).astype(x.dtype)


def layer_norm_c(x, g):
    """Channel LayerNorm (gain only) — module_util.py:70-79.  eps=1e-5 (fp32 branch)."""
    mean = x.mean(axis=1, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=1, keepdims=True)
    return (x - mean) / np.sqrt(var + x.dtype.type(1e-5)) * g


def _ln_st(x, g, res=None):
    """LayerNorm kernel of the engine: normalise (+ residual), then store."""
    y = layer_norm_c(x, g)
    return _st(y if res is None else y + res)


def sinusoidal_pos_emb(t, dim, dtype):
    """SinusoidalPosEmb — module_util.py:29-41.  t: array [b] of ints."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    freqs = np.exp(np.arange(half).astype(np.float32) * np.float32(-e)).astype(np.float32)
    arg = (np.asarray(t).astype(np.float32)[:, None] * freqs[None, :]).astype(np.float32)
    arg = arg.astype(dtype)
    return np.concatenate([np.sin(arg), np.cos(arg)], axis=-1)


def linear(x, w, b):
    return x @ w.T + b


def linear_attention(p, prefix, x, heads=4, dim_head=32):
    """LinearAttention.forward — module_util.py:163-178."""
    B, C, H, W = x.shape
    N = H * W
    qkv = _st(conv2d(x, p[prefix + "to_qkv.weight"]))
    q, k, v = (qkv[:, i * heads * dim_head:(i + 1) * heads * dim_head].reshape(B, heads, dim_head, N)
               for i in range(3))
    q = np.exp(q - q.max(axis=2, keepdims=True))
    q = q / q.sum(axis=2, keepdims=True)
    k = np.exp(k - k.max(axis=3, keepdims=True))
    k = k / k.sum(axis=3, keepdims=True)
    q = q * x.dtype.type(dim_head ** -0.5)
    v = v / x.dtype.type(N)
    context = np.einsum("bhdn,bhen->bhde", k, v)
    out = np.einsum("bhde,bhdn->bhen", context, q)
    out = _st(out.reshape(B, heads * dim_head, H, W))
    out = _st(conv2d(out, p[prefix + "to_out.0.weight"], p[prefix + "to_out.0.bias"]))
    return layer_norm_c(out, p[prefix + "to_out.1.g"])


ATTN_FUSED_16 = False   # restate the engine's fused attention kernels of the 16-bit operand modes (r05, ABI 106) instead of the q | k | v tensor plan


class fused_attn_16:
    """Context manager (inside bf16_convs / f16_convs): Residual(PreNorm(LinearAttention)) blocks with 64 / 128 / 256 channels are restated the way the engine's
    fused kernels compute them in the 16-bit operand modes (attn_block_fused16): no stored q | k | v / attention-output tensor, the roundings at the MFMA operands."""

    def __enter__(self):
        global ATTN_FUSED_16
        self.prev = ATTN_FUSED_16
        ATTN_FUSED_16 = True

    def __exit__(self, *a):
        global ATTN_FUSED_16
        ATTN_FUSED_16 = self.prev


def attn_block_fused16(p, prefix, x, f16=False, store_bf16=False, heads=4, dim_head=32):
    """Residual(PreNorm(LinearAttention)) — module_util.py:20-26,82-90,150-178 — with the roundings of the engine's fused attention kernels in the 16-bit operand modes
    (image_restoration_sde_amd/csrc/kernels_misc.hip: attn_kv_ctx_kernel / attn_q_out_fused_kernel, MODE 2 = bf16, MODE 3 = fp16; r05).  Rounded once, to nearest even:
    the LayerNorm output (operand of the three projections), the to_qkv / to_out weights, exp(k - max) and v (operands of the context sum), the merged context and the
    softmax(q) probabilities (operands of the output product), the attention output (operand of to_out).  Everything else — LayerNorms, both softmaxes, the softmax
    denominator of k (the UNROUNDED exponentials), every accumulation, bias, residual — in the working precision.  fp16: context and attention output pass the
    MFMAs scaled by 2^ceil(log2 N) (exact; undone in front of the bias).  store_bf16: x comes from and the result goes to a bf16 tensor (IRSDE_FLAG_BF16_ACT).
    The kernels take the maximum of k per 128-pixel tile as it runs and merge chunk partials; this restatement uses the global maximum (the difference is rounding
    noise of individual exponentials, far below the bars of the tests)."""
    r = round_f16 if f16 else round_bf16
    B, C, H, W = x.shape
    N = H * W
    hid = heads * dim_head
    xa = r(layer_norm_c(x, p[prefix + "fn.norm.g"])).reshape(B, C, N)
    wqkv = r(p[prefix + "fn.fn.to_qkv.weight"].reshape(3 * hid, C))
    qkv = np.einsum("oc,bcn->bon", wqkv, xa)
    q, k, v = (qkv[:, i * hid:(i + 1) * hid].reshape(B, heads, dim_head, N) for i in range(3))
    ek = np.exp(k - k.max(axis=3, keepdims=True))
    z = ek.sum(axis=3)                                                      # [B, heads, d]: unrounded exponentials
    ctx = np.einsum("bhdn,bhen->bhde", r(ek), r(v)) / z[..., None] / x.dtype.type(N) * x.dtype.type(dim_head ** -0.5)
    pq = np.exp(q - q.max(axis=2, keepdims=True))
    pq = pq / pq.sum(axis=2, keepdims=True)
    up = x.dtype.type(2.0 ** math.ceil(math.log2(N))) if f16 else x.dtype.type(1.0)
    out = np.einsum("bhde,bhdn->bhen", r(ctx * up), r(pq)).reshape(B, hid, N)
    wo = r(p[prefix + "fn.fn.to_out.0.weight"].reshape(C, hid))
    y = np.einsum("co,bon->bcn", wo, r(out)) / up + p[prefix + "fn.fn.to_out.0.bias"].reshape(1, C, 1)
    y = layer_norm_c(y.reshape(B, C, H, W), p[prefix + "fn.fn.to_out.1.g"]) + x
    return round_bf16(y) if store_bf16 else y


def attn_block(p, prefix, x):
    """Residual(PreNorm(dim, LinearAttention(dim))) — module_util.py:20-26,82-90."""
    if ATTN_FUSED_16 and (CONV_OPERANDS_BF16 or CONV_OPERANDS_F16) and x.shape[1] in (64, 128, 256) and (prefix + "fn.fn.to_out.1.g") in p:
        return attn_block_fused16(p, prefix, x, f16=CONV_OPERANDS_F16 and not CONV_OPERANDS_BF16, store_bf16=ACT_STORAGE_BF16)
    return _st(linear_attention(p, prefix + "fn.fn.", _ln_st(x, p[prefix + "fn.norm.g"])) + x)


def attn_sensitive_params(params, H, W, depth=4):
    """Test weights under which the LinearAttention blocks matter (r05).  With default-initialised weights a block's output is dominated by to_out's bias: the
    context carries v / (h w) (module_util.py:168), so to_out(out) is ~1e-6 of the bias in front of the LayerNorm and a wrong attention core moves a network output
    by ~1e-6.  Scaling every block's to_out.0.weight by the pixel count of its level makes the attention branch O(1).  H, W: the input size (padded up to a
    multiple of 2^depth as ConditionalUNet.check_image_size does).  Returns a new dict."""
    s = 2 ** depth
    Hp, Wp = -(-H // s) * s, -(-W // s) * s
    out = dict(params)
    levels = {"downs.%d.2." % i: i for i in range(depth)}
    levels.update({"ups.%d.2." % j: depth - 1 - j for j in range(depth)})
    levels["mid_attn."] = depth - 1
    for pref, lvl in levels.items():
        k = pref + "fn.fn.to_out.0.weight"
        out[k] = (np.asarray(params[k]) * np.asarray(params[k]).dtype.type((Hp >> lvl) * (Wp >> lvl)))
    return out


def res_block(p, prefix, x, temb):
    """ResBlock.forward — module_util.py:136-146 (Block :108-122)."""
    ss = linear(silu(temb), p[prefix + "mlp.1.weight"], p[prefix + "mlp.1.bias"])  # [b, 2C]
    C = ss.shape[1] // 2
    scale = ss[:, :C, None, None]
    shift = ss[:, C:, None, None]
    h = conv2d(x, p[prefix + "block1.proj.weight"], pad=1)
    h = _st(silu(h * (scale + 1) + shift))
    h = silu(conv2d(h, p[prefix + "block2.proj.weight"], pad=1))
    if (prefix + "res_conv.weight") in p:
        return _st(h + _st(conv2d(x, p[prefix + "res_conv.weight"])))
    return _st(h + x)


def upsample_nearest2(x):
    return x.repeat(2, axis=2).repeat(2, axis=3)


def unet_forward(params, xt, cond, t, depth=4, dtype=np.float64, taps=None):
    """ConditionalUNet.forward — DenoisingUNet_arch.py:85-134.

    params: dict of reference state_dict names -> numpy arrays (NCHW / OIHW).
    t: python int (sampling) or int array [B] (training-style).  `taps`: optional dict that
    receives named intermediate activations (NCHW) for per-layer debugging.
    """
    p = {k: np.asarray(v, dtype=dtype) for k, v in params.items()}
    xt = np.asarray(xt, dtype=dtype)
    cond = np.asarray(cond, dtype=dtype)
    if np.isscalar(t):
        t = np.array([int(t)])
    x = np.concatenate([xt - cond, cond], axis=1)  # :90-91
    H, W = x.shape[2:]
    s = 2 ** depth
    ph, pw = (s - H % s) % s, (s - W % s) % s
    x = np.pad(x, ((0, 0), (0, 0), (0, ph), (0, pw)), mode="reflect")  # :78-83
    x = _st(conv2d(x, p["init_conv.weight"], pad=3))  # :96
    x_ = x
    nf = p["init_conv.weight"].shape[0]
    temb = sinusoidal_pos_emb(t, nf, dtype)  # :99
    temb = linear(temb, p["time_mlp.1.weight"], p["time_mlp.1.bias"])
    temb = gelu(temb)
    temb = linear(temb, p["time_mlp.3.weight"], p["time_mlp.3.bias"])

    def tap(name, v):
        if taps is not None:
            taps[name] = v

    tap("init_conv", x)
    tap("time_emb", temb)
    h = []
    for i in range(depth):  # :103-111
        x = res_block(p, "downs.%d.0." % i, x, temb)
        tap("downs.%d.0" % i, x)
        h.append(x)
        x = res_block(p, "downs.%d.1." % i, x, temb)
        tap("downs.%d.1" % i, x)
        x = attn_block(p, "downs.%d.2." % i, x)
        tap("downs.%d.2" % i, x)
        h.append(x)
        if i != depth - 1:
            x = _st(conv2d(x, p["downs.%d.3.weight" % i], p["downs.%d.3.bias" % i], stride=2, pad=1))
        else:
            x = _st(conv2d(x, p["downs.%d.3.weight" % i], pad=1))
        tap("downs.%d.3" % i, x)
    x = res_block(p, "mid_block1.", x, temb)  # :113-115
    tap("mid_block1", x)
    x = attn_block(p, "mid_attn.", x)
    tap("mid_attn", x)
    x = res_block(p, "mid_block2.", x, temb)
    tap("mid_block2", x)
    for j in range(depth):  # :117-125
        x = np.concatenate([x, h.pop()], axis=1)
        x = res_block(p, "ups.%d.0." % j, x, temb)
        tap("ups.%d.0" % j, x)
        x = np.concatenate([x, h.pop()], axis=1)
        x = res_block(p, "ups.%d.1." % j, x, temb)
        tap("ups.%d.1" % j, x)
        x = attn_block(p, "ups.%d.2." % j, x)
        tap("ups.%d.2" % j, x)
        if j != depth - 1:
            x = _st(conv2d(upsample_nearest2(x), p["ups.%d.3.1.weight" % j], p["ups.%d.3.1.bias" % j], pad=1))
        else:
            x = _st(conv2d(x, p["ups.%d.3.weight" % j], pad=1))
        tap("ups.%d.3" % j, x)
    x = np.concatenate([x, x_], axis=1)  # :127
    x = res_block(p, "final_res_block.", x, temb)
    tap("final_res_block", x)
    x = conv2d(x, p["final_conv.weight"], p["final_conv.bias"], pad=1)  # :130
    return np.ascontiguousarray(x[..., :H, :W])  # :132


# --------------------------------------------------------------------------------------
# Parameter inventory + deterministic synthetic weights (shared by golden gen and tests)
# --------------------------------------------------------------------------------------


def unet_param_shapes(in_nc=3, out_nc=3, nf=64, depth=4):
    """Names/shapes of ConditionalUNet's state_dict — DenoisingUNet_arch.py:19-76.
    (151 tensors for nf=64, depth=4; checked against the reference in gen_golden.py.)"""
    sh = {}
    td = nf * 4
    sh["init_conv.weight"] = (nf, in_nc * 2, 7, 7)
    sh["time_mlp.1.weight"] = (td, nf)
    sh["time_mlp.1.bias"] = (td,)
    sh["time_mlp.3.weight"] = (td, td)
    sh["time_mlp.3.bias"] = (td,)

    def resblock(prefix, ci, co):
        sh[prefix + "mlp.1.weight"] = (2 * co, td)
        sh[prefix + "mlp.1.bias"] = (2 * co,)
        sh[prefix + "block1.proj.weight"] = (co, ci, 3, 3)
        sh[prefix + "block2.proj.weight"] = (co, co, 3, 3)
        if ci != co:
            sh[prefix + "res_conv.weight"] = (co, ci, 1, 1)

    def attn(prefix, c):
        sh[prefix + "fn.norm.g"] = (1, c, 1, 1)
        sh[prefix + "fn.fn.to_qkv.weight"] = (384, c, 1, 1)
        sh[prefix + "fn.fn.to_out.0.weight"] = (c, 128, 1, 1)
        sh[prefix + "fn.fn.to_out.0.bias"] = (c,)
        sh[prefix + "fn.fn.to_out.1.g"] = (1, c, 1, 1)

    for i in range(depth):
        di, do = nf * 2 ** i, nf * 2 ** (i + 1)
        resblock("downs.%d.0." % i, di, di)
        resblock("downs.%d.1." % i, di, di)
        attn("downs.%d.2." % i, di)
        if i != depth - 1:
            sh["downs.%d.3.weight" % i] = (do, di, 4, 4)
            sh["downs.%d.3.bias" % i] = (do,)
        else:
            sh["downs.%d.3.weight" % i] = (do, di, 3, 3)
        j = depth - 1 - i
        resblock("ups.%d.0." % j, do + di, do)
        resblock("ups.%d.1." % j, do + di, do)
        attn("ups.%d.2." % j, do)
        if i != 0:
            sh["ups.%d.3.1.weight" % j] = (di, do, 3, 3)
            sh["ups.%d.3.1.bias" % j] = (di,)
        else:
            sh["ups.%d.3.weight" % j] = (di, do, 3, 3)
    mid = nf * 2 ** depth
    resblock("mid_block1.", mid, mid)
    attn("mid_attn.", mid)
    resblock("mid_block2.", mid, mid)
    resblock("final_res_block.", nf * 2, nf)
    sh["final_conv.weight"] = (out_nc, nf, 3, 3)
    sh["final_conv.bias"] = (out_nc,)
    return sh


def synth_params(seed=0, in_nc=3, out_nc=3, nf=64, depth=4, gain=1.0):
    """Deterministic synthetic weights (numpy legacy RandomState: bit-stable across
    machines).  Conv/Linear weights & biases ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) like
    PyTorch's default init; LayerNorm gains ~ U(0.5, 1.5) so the gain path is exercised."""
    rs = np.random.RandomState(seed)
    out = {}
    shapes = unet_param_shapes(in_nc, out_nc, nf, depth)
    for name in sorted(shapes):
        shp = shapes[name]
        if name.endswith(".g"):
            a = rs.uniform(0.5, 1.5, size=shp)
        else:
            if name.endswith("bias"):
                wshape = shapes[name[:-4] + "weight"]
            else:
                wshape = shp
            fan_in = int(np.prod(wshape[1:]))
            bound = gain / math.sqrt(fan_in)
            a = rs.uniform(-bound, bound, size=shp)
        out[name] = a.astype(np.float32)
    return out


def synth_inputs(seed, B, H, W, max_sigma=10.0):
    """LQ in [0,1], x_T = LQ + N(0,1)*max_sigma/255 (IRSDE.noise_state, sde_utils.py:360-361)."""
    rs = np.random.RandomState(seed)
    lq = rs.uniform(0, 1, size=(B, 3, H, W)).astype(np.float32)
    ms = max_sigma / 255 if max_sigma >= 1 else max_sigma
    xT = (lq + rs.standard_normal((B, 3, H, W)).astype(np.float32) * np.float32(ms)).astype(np.float32)
    return lq, xT


def synth_noise(seed, T, shape):
    """Injected per-step N(0,1) draws, index [t] for t in 1..T (index 0 unused)."""
    rs = np.random.RandomState(seed)
    return rs.standard_normal((T + 1,) + tuple(shape)).astype(np.float32)


# --------------------------------------------------------------------------------------
# Sampler loops — sde_utils.py:252-299
# --------------------------------------------------------------------------------------


def sample(params, sch, xT, mu, mode, noise=None, depth=4, dtype=np.float64, T=-1, net=None):
    """IRSDE.reverse_sde / reverse_ode / reverse_posterior (sde_utils.py:252-299) with the
    per-step torch.randn_like replaced by `noise[t]` (shape [T+1,B,3,H,W])."""
    T = sch["T"] if T < 0 else T
    x = np.asarray(xT, dtype=dtype).copy()
    mu = np.asarray(mu, dtype=dtype)
    fwd = net if net is not None else (lambda x_, mu_, t_: unet_forward(params, x_, mu_, t_, depth=depth, dtype=dtype))
    for t in range(T, 0, -1):
        eps_hat = fwd(x, mu, t)
        if mode == "sde":
            x = reverse_sde_step(sch, x, mu, eps_hat, noise[t], t, dtype)
        elif mode == "ode":
            x = reverse_ode_step(sch, x, mu, eps_hat, t, dtype)
        elif mode == "posterior":
            x = reverse_posterior_step(sch, x, mu, eps_hat, noise[t], t, dtype)
        else:
            raise ValueError(mode)
    return x


# --------------------------------------------------------------------------------------
# Philox4x32-10 + Box-Muller: restatement of the device RNG used when no noise is injected
# (new behaviour, not in the reference, which calls torch.randn_like — sde_utils.py:182,223).
# Spec: Salmon et al., "Parallel Random Numbers: As Easy as 1, 2, 3" (SC'11), Philox4x32-10.
# --------------------------------------------------------------------------------------

PHILOX_M0, PHILOX_M1 = 0xD2511F53, 0xCD9E8D57
PHILOX_W0, PHILOX_W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32_10(ctr, key):
    """ctr: uint32 array [...,4]; key: (k0,k1). Returns uint32 array [...,4]."""
    c = [np.asarray(ctr[..., i], dtype=np.uint64) for i in range(4)]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(PHILOX_M0) * c[0]
        p1 = np.uint64(PHILOX_M1) * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c = [(hi1 ^ c[1] ^ k0) & mask, lo1, (hi0 ^ c[3] ^ k1) & mask, lo0]
        k0 = (k0 + np.uint64(PHILOX_W0)) & mask
        k1 = (k1 + np.uint64(PHILOX_W1)) & mask
    return np.stack(c, axis=-1).astype(np.uint32)


def device_normal(seed, t, image_index, n_elems):
    """N(0,1) draws for one image at step t, element order (c,y,x) flattened; matches
    `irsde_philox_normal` in csrc/elementwise.hip: counter=(e//4, t, image, 0x1D5DE),
    key=(seed lo, seed hi); Box-Muller on (r0,r1) and (r2,r3)."""
    nq = (n_elems + 3) // 4
    ctr = np.zeros((nq, 4), dtype=np.uint32)
    ctr[:, 0] = np.arange(nq, dtype=np.uint32)
    ctr[:, 1] = np.uint32(t)
    ctr[:, 2] = np.uint32(image_index)
    ctr[:, 3] = np.uint32(0x1D5DE)
    r = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)).astype(np.float64)
    u = ((np.floor(r / 256.0)) + 0.5) * (1.0 / 16777216.0)  # (r>>8 + 0.5) * 2^-24 in (0,1)
    rad0 = np.sqrt(-2.0 * np.log(u[:, 0]))
    rad1 = np.sqrt(-2.0 * np.log(u[:, 2]))
    a0 = 2.0 * math.pi * u[:, 1]
    a1 = 2.0 * math.pi * u[:, 3]
    z = np.stack([rad0 * np.cos(a0), rad0 * np.sin(a0), rad1 * np.cos(a1), rad1 * np.sin(a1)], axis=1)
    return z.reshape(-1)[:n_elems]


# --------------------------------------------------------------------------------------
# ConditionalNAFNet (Refusion) — codes/config/deraining/models/modules/DenoisingNAFNet_arch.py
# --------------------------------------------------------------------------------------


def naf_param_shapes(img_channel=3, width=64, middle_blk_num=1, enc_blk_nums=(1, 1, 1, 28), dec_blk_nums=(1, 1, 1, 1), lens=False):
    """Names/shapes of ConditionalNAFNet's state_dict — DenoisingNAFNet_arch.py:85-147 (NAFBlock :15-49).  lens=True: the
    latent-bokeh variant (latent-bokeh/models/modules/DenoisingNAFNet_arch.py): time_mlp.{0,2}, cam_mlp.{0,2}, and per block
    time_mlp.1 / cam_mlp.1 instead of mlp.1."""
    sh = {}
    td = width * 4
    t1, t3 = ("time_mlp.0.", "time_mlp.2.") if lens else ("time_mlp.1.", "time_mlp.3.")
    sh[t1 + "weight"] = (td * 2, width)
    sh[t1 + "bias"] = (td * 2,)
    sh[t3 + "weight"] = (td, td)
    sh[t3 + "bias"] = (td,)
    if lens:
        sh["cam_mlp.0.weight"] = (td * 2, width * 3)
        sh["cam_mlp.0.bias"] = (td * 2,)
        sh["cam_mlp.2.weight"] = (td, td)
        sh["cam_mlp.2.bias"] = (td,)
    sh["intro.weight"] = (width, img_channel * 2, 3, 3)
    sh["intro.bias"] = (width,)
    sh["ending.weight"] = (img_channel, width, 3, 3)
    sh["ending.bias"] = (img_channel,)

    def block(p, c):
        tm = "time_mlp.1." if lens else "mlp.1."
        sh[p + tm + "weight"] = (4 * c, td // 2)
        sh[p + tm + "bias"] = (4 * c,)
        if lens:
            sh[p + "cam_mlp.1.weight"] = (2 * c, td // 2)
            sh[p + "cam_mlp.1.bias"] = (2 * c,)
        sh[p + "conv1.weight"] = (2 * c, c, 1, 1)
        sh[p + "conv1.bias"] = (2 * c,)
        sh[p + "conv2.weight"] = (2 * c, 1, 3, 3)
        sh[p + "conv2.bias"] = (2 * c,)
        sh[p + "conv3.weight"] = (c, c, 1, 1)
        sh[p + "conv3.bias"] = (c,)
        sh[p + "sca.1.weight"] = (c, c, 1, 1)
        sh[p + "sca.1.bias"] = (c,)
        sh[p + "conv4.weight"] = (2 * c, c, 1, 1)
        sh[p + "conv4.bias"] = (2 * c,)
        sh[p + "conv5.weight"] = (c, c, 1, 1)
        sh[p + "conv5.bias"] = (c,)
        sh[p + "norm1.g"] = (1, c, 1, 1)
        sh[p + "norm2.g"] = (1, c, 1, 1)
        sh[p + "beta"] = (1, c, 1, 1)
        sh[p + "gamma"] = (1, c, 1, 1)

    chan = width
    for i, num in enumerate(enc_blk_nums):
        for j in range(num):
            block("encoders.%d.%d." % (i, j), chan)
        sh["downs.%d.weight" % i] = (2 * chan, chan, 2, 2)
        sh["downs.%d.bias" % i] = (2 * chan,)
        chan *= 2
    for j in range(middle_blk_num):
        block("middle_blks.%d." % j, chan)
    for i, num in enumerate(dec_blk_nums):
        sh["ups.%d.0.weight" % i] = (chan * 2, chan, 1, 1)
        chan //= 2
        for j in range(num):
            block("decoders.%d.%d." % (i, j), chan)
    return sh


def naf_synth_params(seed=0, **cfg):
    """Deterministic synthetic NAFNet weights.  beta/gamma (zero-initialised in the reference, which would make every
    block the identity) ~ U(-0.5, 0.5); LayerNorm gains ~ U(0.5, 1.5); conv/linear ~ U(+-1/sqrt(fan_in))."""
    rs = np.random.RandomState(seed)
    shapes = naf_param_shapes(**cfg)
    out = {}
    for name in sorted(shapes):
        shp = shapes[name]
        if name.endswith(".g"):
            a = rs.uniform(0.5, 1.5, size=shp)
        elif name.endswith("beta") or name.endswith("gamma"):
            a = rs.uniform(-0.5, 0.5, size=shp)
        else:
            wshape = shapes[name[:-4] + "weight"] if name.endswith("bias") else shp
            bound = 1.0 / math.sqrt(int(np.prod(wshape[1:])))
            a = rs.uniform(-bound, bound, size=shp)
        out[name] = a.astype(np.float32)
    return out


def _dwconv3x3(x, w, b):
    """Depthwise 3x3, pad 1 (nn.Conv2d(groups=C)) — DenoisingNAFNet_arch.py:24-25."""
    B, C, H, W = x.shape
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    out = np.zeros_like(x)
    for ky in range(3):
        for kx in range(3):
            out += xp[:, :, ky:ky + H, kx:kx + W] * w[:, 0, ky, kx].reshape(1, C, 1, 1)
    return out + b.reshape(1, C, 1, 1)


def _simple_gate(x):
    c = x.shape[1] // 2
    return x[:, :c] * x[:, c:]


def naf_block(p, pre, x, temb, cam=None):
    """NAFBlock.forward — DenoisingNAFNet_arch.py:56-82; cam (latent-bokeh :60-91): lens embedding -> FiLM after the FFN gate."""
    half = temb.shape[1] // 2
    tm = "time_mlp.1." if cam is not None else "mlp.1."
    tt = linear(temb[:, :half] * temb[:, half:], p[pre + tm + "weight"], p[pre + tm + "bias"])[:, :, None, None]
    c = x.shape[1]
    shift_att, scale_att, shift_ffn, scale_ffn = (tt[:, i * c:(i + 1) * c] for i in range(4))
    inp = x
    x = layer_norm_c(inp, p[pre + "norm1.g"])
    x = x * (scale_att + 1) + shift_att
    x = conv2d(x, p[pre + "conv1.weight"], p[pre + "conv1.bias"])
    x = _dwconv3x3(x, p[pre + "conv2.weight"], p[pre + "conv2.bias"])
    x = _simple_gate(x)
    pooled = x.mean(axis=(2, 3), keepdims=True)
    x = x * conv2d(pooled, p[pre + "sca.1.weight"], p[pre + "sca.1.bias"])
    x = conv2d(x, p[pre + "conv3.weight"], p[pre + "conv3.bias"])
    y = inp + x * p[pre + "beta"]
    x = layer_norm_c(y, p[pre + "norm2.g"])
    x = x * (scale_ffn + 1) + shift_ffn
    x = conv2d(x, p[pre + "conv4.weight"], p[pre + "conv4.bias"])
    x = _simple_gate(x)
    if cam is not None:
        cc = linear(cam[:, :half] * cam[:, half:], p[pre + "cam_mlp.1.weight"], p[pre + "cam_mlp.1.bias"])[:, :, None, None]
        x = x * (cc[:, :c] + 1) + cc[:, c:]
    x = conv2d(x, p[pre + "conv5.weight"], p[pre + "conv5.bias"])
    return y + x * p[pre + "gamma"]


def _pixel_shuffle2(x):
    B, C, H, W = x.shape
    x = x.reshape(B, C // 4, 2, 2, H, W).transpose(0, 1, 4, 2, 5, 3)
    return np.ascontiguousarray(x.reshape(B, C // 4, 2 * H, 2 * W))


def naf_embeddings(p, t, lens_info=None, dtype=np.float64):
    """The time embedding (and, latent-bokeh, the lens embedding) every NAFBlock receives — the head of
    ConditionalNAFNet.forward, DenoisingNAFNet_arch.py:149-160 (latent-bokeh: its DenoisingNAFNet_arch.py:159-176).
    p: parameters already in `dtype`; returns (temb, cam or None)."""
    if np.isscalar(t):
        t = np.array([int(t)])
    width = p["intro.weight"].shape[0]
    lens = lens_info is not None
    t1, t3 = ("time_mlp.0.", "time_mlp.2.") if lens else ("time_mlp.1.", "time_mlp.3.")
    temb = sinusoidal_pos_emb(t, width, dtype)
    temb = linear(temb, p[t1 + "weight"], p[t1 + "bias"])
    h2 = temb.shape[1] // 2
    temb = linear(temb[:, :h2] * temb[:, h2:], p[t3 + "weight"], p[t3 + "bias"])
    cam = None
    if lens:
        nb = max(len(np.atleast_1d(v)) for v in lens_info)
        emb = np.concatenate([sinusoidal_pos_emb(np.broadcast_to(np.atleast_1d(np.asarray(v, dtype=np.float64)), (nb,)), width, dtype)
                              for v in lens_info], axis=1)
        cam = linear(emb, p["cam_mlp.0.weight"], p["cam_mlp.0.bias"])
        cam = linear(cam[:, :h2] * cam[:, h2:], p["cam_mlp.2.weight"], p["cam_mlp.2.bias"])
    return temb, cam


def nafnet_forward(params, xt, cond, t, enc_blk_nums=(1, 1, 1, 28), middle_blk_num=1, dec_blk_nums=(1, 1, 1, 1),
                   dtype=np.float64, taps=None, intro_skip=False, lens_info=None):
    """ConditionalNAFNet.forward — DenoisingNAFNet_arch.py:149-187.  intro_skip: the latent tasks' variant
    (latent-dehazing/models/modules/DenoisingNAFNet_arch.py:162-176), `ending(x + intro(x))`.  lens_info = [src_lens,
    tgt_lens, disparity] (arrays of 1 or B values): the latent-bokeh variant (its DenoisingNAFNet_arch.py:159-198)."""
    p = {k: np.asarray(v, dtype=dtype) for k, v in params.items()}
    xt = np.asarray(xt, dtype=dtype)
    cond = np.asarray(cond, dtype=dtype)
    x = np.concatenate([xt - cond, cond], axis=1)
    temb, cam = naf_embeddings(p, t, lens_info, dtype)
    B, C, H, W = x.shape
    ps = 2 ** len(enc_blk_nums)
    x = np.pad(x, ((0, 0), (0, 0), (0, (ps - H % ps) % ps), (0, (ps - W % ps) % ps)))  # zero pad (:189-194)
    x = conv2d(x, p["intro.weight"], p["intro.bias"], pad=1)

    def tap(name, v):
        if taps is not None:
            taps[name] = v

    tap("intro", x)
    intro = x
    encs = []
    for i, num in enumerate(enc_blk_nums):
        for j in range(num):
            x = naf_block(p, "encoders.%d.%d." % (i, j), x, temb, cam)
        tap("encoders.%d" % i, x)
        encs.append(x)
        x = conv2d(x, p["downs.%d.weight" % i], p["downs.%d.bias" % i], stride=2, pad=0)
        tap("downs.%d" % i, x)
    for j in range(middle_blk_num):
        x = naf_block(p, "middle_blks.%d." % j, x, temb, cam)
    tap("middle", x)
    for i, num in enumerate(dec_blk_nums):
        x = _pixel_shuffle2(conv2d(x, p["ups.%d.0.weight" % i]))
        x = x + encs[len(encs) - 1 - i]
        tap("ups.%d" % i, x)
        for j in range(num):
            x = naf_block(p, "decoders.%d.%d." % (i, j), x, temb, cam)
        tap("decoders.%d" % i, x)
    if intro_skip:
        x = x + intro
    x = conv2d(x, p["ending.weight"], p["ending.bias"], pad=1)
    return np.ascontiguousarray(x[..., :H, :W])


# --------------------------------------------------------------------------------------
# denoising-sde variant: DenoisingSDE + unconditional UNet with full attention at the bottleneck
#   codes/utils/sde_utils.py:373-593, codes/config/denoising-sde/models/modules/DenoisingUNet_arch.py,
#   module_util.py:182-204 (Attention)
# --------------------------------------------------------------------------------------


def dsde_schedule(max_sigma, T, schedule="cosine"):
    """DenoisingSDE.__init__/_initialize (sde_utils.py:377-426): eps fixed at 0.04, `max_sigma > 1` (not >=) rescales."""
    ms = max_sigma / 255 if max_sigma > 1 else max_sigma
    sch = irsde_schedule(ms if ms < 1 else ms, T, schedule if schedule == "cosine" else "linear", eps=0.04)
    sch["max_sigma"] = ms
    # irsde_schedule divides by 255 when max_sigma >= 1; ms is already < 1 for every real config
    return sch


def dsde_reverse_step(sch, x, noise, z, t, ode, dtype=np.float32):
    """SDE.reverse_sde_step / reverse_ode_step with DenoisingSDE's drifts (sde_utils.py:44-48, 448-457)."""
    d = dtype
    x, noise = np.asarray(x, dtype=d), np.asarray(noise, dtype=d)
    sigma, sbar, dt = d(sch["sigmas"][t]), d(sch["sigma_bars"][t]), d(sch["dt"])
    A = d(np.exp(np.float32(-2) * sch["thetas_cumsum"][t] * sch["dt"], dtype=np.float32))
    score = -noise / sbar
    if ode:
        return (x - d(-0.5) * sigma ** 2 * A * score * dt).astype(d)
    drift = d(-0.5) * sigma ** 2 * (1 + A) * score * dt
    disp = sigma * (np.asarray(z, dtype=d) * d(math.sqrt(float(sch["dt"]))))
    return (x - drift - disp).astype(d)


def dsde_optimal_timestep(sch, sigma, eps=1e-6):
    """DenoisingSDE.get_optimal_timestep (sde_utils.py:547-551)."""
    sigma = sigma / 255 if sigma > 1 else sigma
    hat = -1 / (2 * float(sch["dt"])) * math.log(1 - sigma ** 2 / sch["max_sigma"] ** 2 + eps)
    return int(np.argmin(np.abs(sch["thetas_cumsum"] - np.float32(hat))))


def uncond_unet_param_shapes(in_nc=3, out_nc=3, nf=64, depth=4):
    """state_dict of the denoising-sde ConditionalUNet: init_conv takes in_nc channels, mid_attn is full Attention
    (to_out is a bare Conv2d: no LayerNorm) — denoising-sde/.../DenoisingUNet_arch.py:26,71."""
    sh = unet_param_shapes(in_nc, out_nc, nf, depth)
    sh["init_conv.weight"] = (nf, in_nc, 7, 7)
    mid = nf * 2 ** depth
    for k in ("mid_attn.fn.fn.to_out.0.weight", "mid_attn.fn.fn.to_out.0.bias", "mid_attn.fn.fn.to_out.1.g"):
        del sh[k]
    sh["mid_attn.fn.fn.to_out.weight"] = (mid, 128, 1, 1)
    sh["mid_attn.fn.fn.to_out.bias"] = (mid,)
    return sh


def uncond_synth_params(seed=0, in_nc=3, out_nc=3, nf=64, depth=4):
    rs = np.random.RandomState(seed)
    shapes = uncond_unet_param_shapes(in_nc, out_nc, nf, depth)
    out = {}
    for name in sorted(shapes):
        shp = shapes[name]
        if name.endswith(".g"):
            a = rs.uniform(0.5, 1.5, size=shp)
        else:
            wshape = shapes[name[:-4] + "weight"] if name.endswith("bias") else shp
            bound = 1.0 / math.sqrt(int(np.prod(wshape[1:])))
            a = rs.uniform(-bound, bound, size=shp)
        out[name] = a.astype(np.float32)
    return out


def full_attention(p, prefix, x, heads=4, dim_head=32):
    """Attention.forward — module_util.py:193-204."""
    B, C, H, W = x.shape
    N = H * W
    qkv = _st(conv2d(x, p[prefix + "to_qkv.weight"]))
    q, k, v = (qkv[:, i * heads * dim_head:(i + 1) * heads * dim_head].reshape(B, heads, dim_head, N) for i in range(3))
    q = q * x.dtype.type(dim_head ** -0.5)
    sim = np.einsum("bhdi,bhdj->bhij", q, k)
    sim = sim - sim.max(axis=-1, keepdims=True)
    attn = np.exp(sim)
    attn = attn / attn.sum(axis=-1, keepdims=True)
    out = np.einsum("bhij,bhdj->bhid", attn, v)  # b h N d
    out = out.transpose(0, 1, 3, 2).reshape(B, heads * dim_head, H, W)
    return conv2d(out, p[prefix + "to_out.weight"], p[prefix + "to_out.bias"])


def uncond_unet_forward(params, x, t, depth=4, dtype=np.float64, taps=None):
    """denoising-sde ConditionalUNet.forward(x, time) — denoising-sde/.../DenoisingUNet_arch.py:84-130."""
    p = {k: np.asarray(v, dtype=dtype) for k, v in params.items()}
    x = np.asarray(x, dtype=dtype)
    if np.isscalar(t):
        t = np.array([int(t)])
    H, W = x.shape[2:]
    s = 2 ** depth
    x = np.pad(x, ((0, 0), (0, 0), (0, (s - H % s) % s), (0, (s - W % s) % s)), mode="reflect")
    x = conv2d(x, p["init_conv.weight"], pad=3)
    x_ = x
    nf = p["init_conv.weight"].shape[0]
    temb = sinusoidal_pos_emb(t, nf, dtype)
    temb = linear(gelu(linear(temb, p["time_mlp.1.weight"], p["time_mlp.1.bias"])), p["time_mlp.3.weight"], p["time_mlp.3.bias"])

    def tap(name, v):
        if taps is not None:
            taps[name] = v

    h = []
    for i in range(depth):
        x = res_block(p, "downs.%d.0." % i, x, temb)
        h.append(x)
        x = res_block(p, "downs.%d.1." % i, x, temb)
        x = attn_block(p, "downs.%d.2." % i, x)
        h.append(x)
        if i != depth - 1:
            x = conv2d(x, p["downs.%d.3.weight" % i], p["downs.%d.3.bias" % i], stride=2, pad=1)
        else:
            x = conv2d(x, p["downs.%d.3.weight" % i], pad=1)
    x = res_block(p, "mid_block1.", x, temb)
    tap("mid_block1", x)
    x = full_attention(p, "mid_attn.fn.fn.", layer_norm_c(x, p["mid_attn.fn.norm.g"])) + x
    tap("mid_attn", x)
    x = res_block(p, "mid_block2.", x, temb)
    for j in range(depth):
        x = res_block(p, "ups.%d.0." % j, np.concatenate([x, h.pop()], axis=1), temb)
        x = res_block(p, "ups.%d.1." % j, np.concatenate([x, h.pop()], axis=1), temb)
        x = attn_block(p, "ups.%d.2." % j, x)
        if j != depth - 1:
            x = conv2d(upsample_nearest2(x), p["ups.%d.3.1.weight" % j], p["ups.%d.3.1.bias" % j], pad=1)
        else:
            x = conv2d(x, p["ups.%d.3.weight" % j], pad=1)
    x = res_block(p, "final_res_block.", np.concatenate([x, x_], axis=1), temb)
    x = conv2d(x, p["final_conv.weight"], p["final_conv.bias"], pad=1)
    return np.ascontiguousarray(x[..., :H, :W])


def dsde_sample(params, sch, xT, ode, noise=None, depth=4, dtype=np.float64, T=-1):
    """DenoisingSDE.reverse_sde / reverse_ode (sde_utils.py:488-528) with injected noise."""
    T = sch["T"] if T < 0 else T
    x = np.asarray(xT, dtype=dtype).copy()
    for t in range(T, 0, -1):
        eps_hat = uncond_unet_forward(params, x, t, depth=depth, dtype=dtype)
        x = dsde_reverse_step(sch, x, eps_hat, None if ode else noise[t], t, ode, dtype)
    return x


# ---------------------------------------------------------------------------------------------
# Evaluation tail: codes/utils/img_utils.py:136-234, codes/data/util.py:177-198,
# used as in codes/config/deraining/test.py:110-178.  Images are numpy HWC BGR (or HW), as in the reference.
# ---------------------------------------------------------------------------------------------
def tensor2img(t):
    """img_utils.py:136-163 for a (C,H,W) or (H,W) float array: clamp, *255 in float32, round half to even, uint8,
    RGB -> BGR, HWC."""
    t = np.clip(np.asarray(t, np.float32), 0.0, 1.0)
    if t.ndim == 3:
        t = t[::-1].transpose(1, 2, 0) if t.shape[0] == 3 else t[0]
    return np.round(t * np.float32(255.0)).astype(np.uint8)


def calculate_psnr(img1, img2):
    """img_utils.py:182-189."""
    mse = np.mean((img1.astype(np.float64) - img2.astype(np.float64)) ** 2)
    return float("inf") if mse == 0 else 20 * math.log10(255.0 / math.sqrt(mse))


def ssim_plane(a, b):
    """img_utils.py:192-214 for one 2-D plane: 11x11 Gaussian (sigma 1.5) window, 'valid' region, float64."""
    a, b = a.astype(np.float64), b.astype(np.float64)
    i = np.arange(11, dtype=np.float64) - 5.0
    g = np.exp(-(i * i) / (2.0 * 1.5 * 1.5))
    g /= g.sum()
    win = np.outer(g, g)

    def filt(x):
        v = np.lib.stride_tricks.sliding_window_view(x, (11, 11))
        return np.einsum("yxij,ij->yx", v, win)

    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    mu1, mu2 = filt(a), filt(b)
    s11, s22, s12 = filt(a * a) - mu1 ** 2, filt(b * b) - mu2 ** 2, filt(a * b) - mu1 * mu2
    return ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 ** 2 + mu2 ** 2 + C1) * (s11 + s22 + C2))


def calculate_ssim(img1, img2):
    """img_utils.py:217-234: a 3-channel image is filtered per channel and the map is averaged over everything
    (the reference's loop calls ssim() on the whole image three times and averages three equal numbers)."""
    if img1.ndim == 2:
        return float(ssim_plane(img1, img2).mean())
    return float(np.mean([ssim_plane(img1[..., c], img2[..., c]) for c in range(img1.shape[2])]))


def bgr2ycbcr_y(img01):
    """data/util.py:177-198 (only_y=True) for a float image in [0,1], BGR HWC; returns Y in [0,1]."""
    img = img01.astype(np.float64) * 255.0
    return (np.dot(img, [24.966, 128.553, 65.481]) / 255.0 + 16.0) / 255.0


def eval_tail(out_chw, gt_chw, crop_border=0):
    """deraining/test.py:110-178 for one image: (psnr, ssim, psnr_y, ssim_y)."""
    o, g = tensor2img(out_chw), tensor2img(gt_chw)
    sr, gt = o / 255.0, g / 255.0
    cb = crop_border
    crop = (lambda im: im) if cb == 0 else (lambda im: im[cb:-cb, cb:-cb])
    res = [calculate_psnr(crop(sr) * 255, crop(gt) * 255), calculate_ssim(crop(sr) * 255, crop(gt) * 255)]
    if o.ndim == 3:
        sy, gy = bgr2ycbcr_y(sr), bgr2ycbcr_y(gt)
        res += [calculate_psnr(crop(sy) * 255, crop(gy) * 255), calculate_ssim(crop(sy) * 255, crop(gy) * 255)]
    else:
        res += [float("nan"), float("nan")]
    return res


# ---------------------------------------------------------------------------------------------
# Latent wrapper: codes/config/latent-dehazing/models/modules/UNet_arch.py
# ---------------------------------------------------------------------------------------------
def latent_unet_param_shapes(in_ch=3, out_ch=3, ch=8, ch_mult=(4, 8, 8, 16), embed_dim=8):
    """state_dict inventory of UNet(in_ch, out_ch, ch, ch_mult, embed_dim) — UNet_arch.py:18-50 (69 tensors for nasde.yml)."""
    depth = len(ch_mult)
    mult = [1] + list(ch_mult)
    s = {"init_conv.weight": (ch, in_ch, 3, 3)}

    def resb(pre, ci, co):
        s[pre + "block1.proj.weight"] = (co, ci, 3, 3)
        s[pre + "block2.proj.weight"] = (co, co, 3, 3)
        if ci != co:
            s[pre + "res_conv.weight"] = (co, ci, 1, 1)

    def attn(pre, c):
        s[pre + "fn.norm.g"] = (1, c, 1, 1)
        s[pre + "fn.fn.to_qkv.weight"] = (384, c, 1, 1)
        s[pre + "fn.fn.to_out.0.weight"] = (c, 128, 1, 1)
        s[pre + "fn.fn.to_out.0.bias"] = (c,)
        s[pre + "fn.fn.to_out.1.g"] = (1, c, 1, 1)

    for i in range(depth):
        di, do = ch * mult[i], ch * mult[i + 1]
        e, d = "encoder.%d." % i, "decoder.%d." % (depth - 1 - i)
        resb(e + "0.", di, di)
        resb(e + "1.", di, di)
        resb(d + "0.", do + di, do)
        resb(d + "1.", do + di, do)
        if i == depth - 1:
            attn(e + "2.", di)
            attn(d + "2.", do)
            s[e + "3.weight"] = (do, di, 3, 3)
        else:
            s[e + "3.weight"] = (do, di, 4, 4)
            s[e + "3.bias"] = (do,)
        if i != 0:
            s[d + "3.1.weight"] = (di, do, 3, 3)
            s[d + "3.1.bias"] = (di,)
        else:
            s[d + "3.weight"] = (di, do, 3, 3)
    mid = ch * mult[-1]
    s["latent_conv.weight"] = (embed_dim, mid, 1, 1)
    s["post_latent_conv.weight"] = (mid, embed_dim, 1, 1)
    s["final_conv.weight"] = (out_ch, ch, 3, 3)
    s["final_conv.bias"] = (out_ch,)
    return s


def latent_unet_synth_params(seed=0, **cfg):
    rs = np.random.RandomState(seed)
    shapes = latent_unet_param_shapes(**cfg)
    out = {}
    for name in sorted(shapes):
        shp = shapes[name]
        if name.endswith(".g"):
            a = rs.uniform(0.5, 1.5, size=shp)
        else:
            wshape = shapes[name[:-4] + "weight"] if name.endswith("bias") else shp
            bound = 1.0 / math.sqrt(int(np.prod(wshape[1:])))
            a = rs.uniform(-bound, bound, size=shp)
        out[name] = a.astype(np.float32)
    return out


def _plain_res_block(p, pre, x):
    """ResBlock without a time MLP (latent-dehazing module_util.py:132-153): conv -> SiLU -> conv -> SiLU, + res_conv(x)."""
    h = silu(conv2d(x, p[pre + "block1.proj.weight"], pad=1))
    h = silu(conv2d(h, p[pre + "block2.proj.weight"], pad=1))
    return h + (conv2d(x, p[pre + "res_conv.weight"]) if (pre + "res_conv.weight") in p else x)


def latent_unet_encode(params, x, depth, dtype=np.float64):
    """UNet.encode — UNet_arch.py:59-77: returns (latent, hidden list)."""
    p = {k: np.asarray(v, dtype=dtype) for k, v in params.items()}
    x = np.asarray(x, dtype=dtype)
    H, W = x.shape[2:]
    s = 2 ** depth
    x = np.pad(x, ((0, 0), (0, 0), (0, (s - H % s) % s), (0, (s - W % s) % s)), mode="reflect")
    x = conv2d(x, p["init_conv.weight"], pad=1)
    h = [x]
    for i in range(depth):
        e = "encoder.%d." % i
        x = _plain_res_block(p, e + "0.", x)
        h.append(x)
        x = _plain_res_block(p, e + "1.", x)
        if i == depth - 1:
            x = attn_block(p, e + "2.", x)
        h.append(x)
        if i != depth - 1:
            x = conv2d(x, p[e + "3.weight"], p[e + "3.bias"], stride=2, pad=1)
        else:
            x = conv2d(x, p[e + "3.weight"], pad=1)
    return conv2d(x, p["latent_conv.weight"]), h


def latent_unet_decode(params, x, h, depth, H, W, dtype=np.float64):
    """UNet.decode — UNet_arch.py:79-91."""
    p = {k: np.asarray(v, dtype=dtype) for k, v in params.items()}
    x = conv2d(np.asarray(x, dtype=dtype), p["post_latent_conv.weight"])
    h = [np.asarray(t, dtype=dtype) for t in h]
    for i in range(depth):
        d = "decoder.%d." % i
        x = _plain_res_block(p, d + "0.", np.concatenate([x, h[-(i * 2 + 1)]], axis=1))
        x = _plain_res_block(p, d + "1.", np.concatenate([x, h[-(i * 2 + 2)]], axis=1))
        if i == 0:
            x = attn_block(p, d + "2.", x)
        if i != depth - 1:
            x = conv2d(upsample_nearest2(x), p[d + "3.1.weight"], p[d + "3.1.bias"], pad=1)
        else:
            x = conv2d(x, p[d + "3.weight"], pad=1)
    x = conv2d(x + h[0], p["final_conv.weight"], p["final_conv.bias"], pad=1)
    return np.ascontiguousarray(x[..., :H, :W])
