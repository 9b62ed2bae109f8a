"""`ConditionalUNet` — the reference's score-network interface on top of the HIP engine.

Mirrors `ConditionalUNet(in_nc, out_nc, nf, depth=4, upscale=1)` and `forward(xt, cond, time)` of
/root/reference/codes/config/deraining/models/modules/DenoisingUNet_arch.py:18-134 so that
`getattr(models.modules, "ConditionalUNet")(**setting)` (deraining/models/networks.py:10-15) and
`load_state_dict` of a reference checkpoint (151 tensors) work unchanged.

The sub-modules below only OWN parameters under the reference's state_dict names (and give them
PyTorch's default initialisation); none of their `forward`s is ever called.  All arithmetic runs in
libirsde_hip.so (csrc/): `forward` hands raw device pointers to `irsde_unet_forward`.  There is no
PyTorch fallback — a CPU tensor or a missing library raises.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib


class _Block(nn.Module):  # module_util.py:108-112 (parameter container)
    def __init__(self, ci, co):
        super().__init__()
        self.proj = nn.Conv2d(ci, co, 3, padding=1, bias=False)


class _ResBlock(nn.Module):  # module_util.py:125-134 (parameter container)
    def __init__(self, ci, co, time_dim):
        super().__init__()
        self.mlp = nn.Sequential(nn.SiLU(), nn.Linear(time_dim, co * 2))
        self.block1 = _Block(ci, co)
        self.block2 = _Block(co, co)
        self.res_conv = nn.Conv2d(ci, co, 1, bias=False) if ci != co else nn.Identity()


class _Gain(nn.Module):  # module_util.py:70-73
    def __init__(self, dim):
        super().__init__()
        self.g = nn.Parameter(torch.ones(1, dim, 1, 1))


class _LinearAttention(nn.Module):  # module_util.py:150-161
    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        hidden = heads * dim_head
        self.to_qkv = nn.Conv2d(dim, hidden * 3, 1, bias=False)
        self.to_out = nn.Sequential(nn.Conv2d(hidden, dim, 1), _Gain(dim))


class _PreNorm(nn.Module):  # module_util.py:82-86
    def __init__(self, dim):
        super().__init__()
        self.fn = _LinearAttention(dim)
        self.norm = _Gain(dim)


class _Residual(nn.Module):  # module_util.py:20-23
    def __init__(self, dim):
        super().__init__()
        self.fn = _PreNorm(dim)


def _upsample(dim, dim_out):  # module_util.py:93-97: index 1 of the Sequential holds the conv
    return nn.Sequential(nn.Identity(), nn.Conv2d(dim, dim_out, 3, 1, 1))


class _Engine:
    """Owns one irsde_engine handle (weights are immutable once finalized)."""

    def __init__(self, module, device_index, flags=0):
        L = _lib.lib()
        self.h = module._create_handle(L, device_index, flags)
        h = self.h
        self.schedule_key = None
        sd = {k: v.detach().to("cpu", torch.float32).contiguous() for k, v in module.state_dict().items()}
        n = L.irsde_num_weights(h)
        names = [L.irsde_weight_name(h, i).decode() for i in range(n)]
        if set(names) != set(sd.keys()):
            raise _lib.IrsdeError("state_dict / engine inventory mismatch: %s" % sorted(set(names) ^ set(sd)))
        for name in names:
            t = sd[name]
            shape = (ctypes.c_int64 * t.dim())(*t.shape)
            _lib.check(L.irsde_load_weight(h, name.encode(), ctypes.c_void_p(t.data_ptr()), shape, t.dim()))
        with torch.cuda.device(device_index):
            _lib.check(L.irsde_finalize_weights(h))

    def __del__(self):
        try:
            if self.h:
                _lib.lib().irsde_destroy(self.h)
                self.h = None
        except Exception:
            pass


class ConditionalUNet(nn.Module):
    def __init__(self, in_nc, out_nc, nf, depth=4, upscale=1):
        super().__init__()
        self.in_nc, self.out_nc, self.nf, self.depth = in_nc, out_nc, nf, depth
        self.upscale = upscale  # unused, as in the reference (DenoisingUNet_arch.py:23)
        time_dim = nf * 4
        self.init_conv = nn.Conv2d(in_nc * 2, nf, 7, padding=3, bias=False)
        self.time_mlp = nn.Sequential(nn.Identity(), nn.Linear(nf, time_dim), nn.GELU(), nn.Linear(time_dim, time_dim))
        self.downs = nn.ModuleList([])
        self.ups = nn.ModuleList([])
        for i in range(depth):
            di, do = nf * 2 ** i, nf * 2 ** (i + 1)
            self.downs.append(nn.ModuleList([
                _ResBlock(di, di, time_dim), _ResBlock(di, di, time_dim), _Residual(di),
                nn.Conv2d(di, do, 4, 2, 1) if i != depth - 1 else nn.Conv2d(di, do, 3, padding=1, bias=False)]))
            self.ups.insert(0, nn.ModuleList([
                _ResBlock(do + di, do, time_dim), _ResBlock(do + di, do, time_dim), _Residual(do),
                _upsample(do, di) if i != 0 else nn.Conv2d(do, di, 3, padding=1, bias=False)]))
        mid = nf * 2 ** depth
        self.mid_block1 = _ResBlock(mid, mid, time_dim)
        self.mid_attn = _Residual(mid)
        self.mid_block2 = _ResBlock(mid, mid, time_dim)
        self.final_res_block = _ResBlock(nf * 2, nf, time_dim)
        self.final_conv = nn.Conv2d(nf, out_nc, 3, 1, 1)
        self._engine = None
        self._engine_key = None
        self.engine_flags = 0

    # ---- engine management -------------------------------------------------------------------
    def _create_handle(self, L, device_index, flags):
        cfg = _lib.Config(self.in_nc, self.out_nc, self.nf, self.depth, device_index, flags)
        h = ctypes.c_void_p()
        _lib.check(L.irsde_create(ctypes.byref(cfg), ctypes.byref(h)))
        return h

    def _param_key(self, device):
        return (device.index if device.index is not None else torch.cuda.current_device(),
                self.engine_flags, tuple((p.data_ptr(), p._version) for p in self.parameters()))

    def engine(self, device=None):
        """The HIP engine holding the current parameter values (rebuilt if they changed)."""
        if device is None:
            device = next(self.parameters()).device
        if device.type != "cuda":
            raise _lib.IrsdeError("ConditionalUNet runs only on an AMD GPU through libirsde_hip.so "
                                  "(no CPU/PyTorch fallback); move the module to 'cuda' first")
        key = self._param_key(device)
        if self._engine is None or self._engine_key != key:
            self._engine = _Engine(self, key[0], self.engine_flags)
            self._engine_key = key
        return self._engine

    def set_compute_dtype(self, dtype):
        """'fp32' (default: exact-fp32 MFMA + Winograd), 'bf16' (BASELINE configs[2]: conv operands rounded to bf16,
        fp32 accumulation, everything else fp32), 'bf16_act' (+ bf16 activation storage) or 'fp16' (BASELINE configs[4]:
        the 'bf16' mode with IEEE fp16 operands), or 'fp32_split' (r03: fp32 everywhere, but the deep Winograd component GEMMs
        multiply bf16 hi + lo pairs of their fp32 operands on the bf16 MFMA pipe, IRSDE_FLAG_SPLIT_BF16X2; 'fp32_split_f16': fp16 pairs).  New behaviour — the
        reference is fp32 only."""
        self.engine_flags &= ~(_lib.FLAG_BF16 | _lib.FLAG_BF16_ACT | _lib.FLAG_FP16 | _lib.FLAG_SPLIT_BF16X2 | _lib.FLAG_SPLIT_F16X2)
        if dtype in ("fp32", "f32", torch.float32):
            pass
        elif dtype == "fp32_split":
            self.engine_flags |= _lib.FLAG_SPLIT_BF16X2
        elif dtype == "fp32_split_f16":   # the same path with fp16 hi + lo pieces: fp32-equivalent per layer, range-limited (|activation| < ~1e4)
            self.engine_flags |= _lib.FLAG_SPLIT_F16X2
        elif dtype in ("bf16", torch.bfloat16):
            self.engine_flags |= _lib.FLAG_BF16
        elif dtype == "bf16_act":  # + bf16 storage of the activation tensors (conditional UNet only)
            self.engine_flags |= _lib.FLAG_BF16 | _lib.FLAG_BF16_ACT
        elif dtype in ("fp16", "f16", torch.float16):
            self.engine_flags |= _lib.FLAG_FP16
        else:
            raise _lib.IrsdeError("compute dtype must be 'fp32', 'fp32_split', 'fp32_split_f16', 'bf16', 'bf16_act' or 'fp16'")
        return self

    # ---- reference interface -----------------------------------------------------------------
    def forward(self, xt, cond, time):
        """noise = model(xt, cond, time) — DenoisingUNet_arch.py:85-134."""
        if isinstance(time, (int, float)):
            ts = [int(time)]
        else:
            ts = [int(v) for v in torch.as_tensor(time).reshape(-1).tolist()]
        if xt.device.type != "cuda":
            raise _lib.IrsdeError("ConditionalUNet.forward needs CUDA(HIP) tensors; got %s" % xt.device)
        B, C, H, W = xt.shape
        if len(ts) not in (1, B):
            raise _lib.IrsdeError("time must hold 1 or B timesteps")
        eng = self.engine(xt.device)
        x = xt.detach().to(torch.float32).contiguous()
        c = cond.detach().to(torch.float32).contiguous()
        out = torch.empty((B, self.out_nc, H, W), device=xt.device, dtype=torch.float32)
        tarr = (ctypes.c_int64 * len(ts))(*ts)
        with torch.cuda.device(xt.device):
            _lib.check(_lib.lib().irsde_unet_forward(
                eng.h, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(c.data_ptr()), tarr, len(ts), B, H, W,
                ctypes.c_void_p(out.data_ptr()), _lib.stream_ptr()))
        return out

    def debug_tap(self, name):
        """Named intermediate activation (NCHW, host) of the last forward; needs engine_flags |= 1."""
        eng = self.engine()
        dims = (ctypes.c_int64 * 4)()
        _lib.check(_lib.lib().irsde_debug_tap(eng.h, name.encode(), None, dims))
        out = torch.empty(tuple(dims), dtype=torch.float32)
        _lib.check(_lib.lib().irsde_debug_tap(eng.h, name.encode(), ctypes.c_void_p(out.data_ptr()), dims))
        return out
