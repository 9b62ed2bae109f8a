"""CPU port of the reference sampler in torch.nn.functional ops (TEST INFRASTRUCTURE / CPU BASELINE ONLY).

Same algorithm as oracle/irsde_oracle.py (which is the numpy checker), written with the torch CPU ops the
reference itself dispatches to (F.conv2d / F.linear / softmax / einsum: oneDNN + ATen on the host cores),
so that `bench.py`'s `cpu_baseline` leg times what the reference's CPU path would cost on the GPU box's
host.  /root/reference is not available there, hence this port ("kind": "port").  Pinned against the real
reference's golden vectors in tests/test_oracle_golden.py.  Never imported by the product package.

Reference lines: DenoisingUNet_arch.py:85-134, module_util.py:70-178, sde_utils.py:175-190,252-266.
"""
import math

import torch
import torch.nn.functional as F


def _ln(x, g):  # module_util.py:74-79
    var = torch.var(x, dim=1, unbiased=False, keepdim=True)
    mean = torch.mean(x, dim=1, keepdim=True)
    return (x - mean) * (var + 1e-5).rsqrt() * g


def _res_block(p, pre, x, temb):  # module_util.py:136-146
    ss = F.linear(F.silu(temb), p[pre + "mlp.1.weight"], p[pre + "mlp.1.bias"])[:, :, None, None]
    scale, shift = ss.chunk(2, dim=1)
    h = F.silu(F.conv2d(x, p[pre + "block1.proj.weight"], padding=1) * (scale + 1) + shift)
    h = F.silu(F.conv2d(h, p[pre + "block2.proj.weight"], padding=1))
    if pre + "res_conv.weight" in p:
        return h + F.conv2d(x, p[pre + "res_conv.weight"])
    return h + x


def _attn(p, pre, x, heads=4, dh=32):  # module_util.py:20-26,82-90,163-178
    b, c, h, w = x.shape
    xn = _ln(x, p[pre + "fn.norm.g"])
    q, k, v = F.conv2d(xn, p[pre + "fn.fn.to_qkv.weight"]).chunk(3, dim=1)
    q, k, v = (t.reshape(b, heads, dh, h * w) for t in (q, k, v))
    q = q.softmax(dim=-2) * dh ** -0.5
    k = k.softmax(dim=-1)
    v = v / (h * w)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(b, heads * dh, h, w)
    out = F.conv2d(out, p[pre + "fn.fn.to_out.0.weight"], p[pre + "fn.fn.to_out.0.bias"])
    return _ln(out, p[pre + "fn.fn.to_out.1.g"]) + x


def unet_forward(p, xt, cond, t, depth=4):
    """p: dict name -> torch CPU tensor (reference state_dict layout)."""
    if isinstance(t, (int, float)):
        t = torch.tensor([t])
    x = torch.cat([xt - cond, cond], dim=1)
    H, W = x.shape[2:]
    s = 2 ** depth
    x = F.pad(x, (0, (s - W % s) % s, 0, (s - H % s) % s), "reflect")
    x = F.conv2d(x, p["init_conv.weight"], padding=3)
    x_ = x
    nf = p["init_conv.weight"].shape[0]
    half = nf // 2
    freqs = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
    emb = t[:, None] * freqs[None, :]
    emb = torch.cat((emb.sin(), emb.cos()), dim=-1)
    temb = F.linear(F.gelu(F.linear(emb, p["time_mlp.1.weight"], p["time_mlp.1.bias"])),
                    p["time_mlp.3.weight"], p["time_mlp.3.bias"])
    hs = []
    for i in range(depth):
        d = "downs.%d." % i
        x = _res_block(p, d + "0.", x, temb)
        hs.append(x)
        x = _res_block(p, d + "1.", x, temb)
        x = _attn(p, d + "2.", x)
        hs.append(x)
        if i != depth - 1:
            x = F.conv2d(x, p[d + "3.weight"], p[d + "3.bias"], stride=2, padding=1)
        else:
            x = F.conv2d(x, p[d + "3.weight"], padding=1)
    x = _res_block(p, "mid_block1.", x, temb)
    x = _attn(p, "mid_attn.", x)
    x = _res_block(p, "mid_block2.", x, temb)
    for j in range(depth):
        u = "ups.%d." % j
        x = _res_block(p, u + "0.", torch.cat([x, hs.pop()], dim=1), temb)
        x = _res_block(p, u + "1.", torch.cat([x, hs.pop()], dim=1), temb)
        x = _attn(p, u + "2.", x)
        if j != depth - 1:
            x = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), p[u + "3.1.weight"], p[u + "3.1.bias"],
                         padding=1)
        else:
            x = F.conv2d(x, p[u + "3.weight"], padding=1)
    x = _res_block(p, "final_res_block.", torch.cat([x, x_], dim=1), temb)
    x = F.conv2d(x, p["final_conv.weight"], p["final_conv.bias"], padding=1)
    return x[..., :H, :W]


def reverse_sde_steps(p, sch, x, mu, noise, t_from, n_steps, depth=4):
    """n_steps of IRSDE.reverse_sde (sde_utils.py:252-266) starting at t_from; sch = numpy tables."""
    with torch.no_grad():
        for t in range(t_from, t_from - n_steps, -1):
            eps_hat = unet_forward(p, x, mu, t, depth)
            score = -eps_hat / float(sch["sigma_bars"][t])
            drift = (float(sch["thetas"][t]) * (mu - x) - float(sch["sigmas"][t]) ** 2 * score) * float(sch["dt"])
            x = x - drift - float(sch["sigmas"][t]) * (noise[t] * math.sqrt(float(sch["dt"])))
    return x
