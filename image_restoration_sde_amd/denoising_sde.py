"""The denoising-sde task variant on the HIP engine:

  `DenoisingSDE`     /root/reference/codes/utils/sde_utils.py:373-593 — the mu-free SDE for Gaussian denoising
  `ConditionalUNet`  /root/reference/codes/config/denoising-sde/models/modules/DenoisingUNet_arch.py:19-130 — same class
                     name as the reference's (that task directory resolves `which_model_G: ConditionalUNet` inside its
                     own `models.modules`): `forward(x, time)`, no condition input, full softmax `Attention` at the
                     bottleneck (module_util.py:182-204) — the one place the N x N QK^T / AV contractions exist; they run
                     as a flash-style fp32-MFMA kernel (csrc/kernels_misc.hip: full_attn_kernel).

`DenoisingModel.test(sde, sigma)` of that task (denoising-sde/models/denoising_model.py:162-170) calls
`sde.reverse_ode(LQ, T=sde.get_optimal_timestep(sigma))`; both reverse samplers run entirely inside libirsde_hip.so when
the model is this module's `ConditionalUNet`, and fall back to the per-step fused HIP update for foreign models.
"""
import ctypes
import math

import torch
import torch.nn as nn

from . import _lib
from .sde import _check_fp16_range, _save_state, _unwrap
from .unet import ConditionalUNet as _CondUNet, _Gain, _ResBlock, _Residual, _upsample


class _FullAttention(nn.Module):  # module_util.py:182-191 (parameter container)
    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        self.to_qkv = nn.Conv2d(dim, heads * dim_head * 3, 1, bias=False)
        self.to_out = nn.Conv2d(heads * dim_head, dim, 1)


class _PreNormFull(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.fn = _FullAttention(dim)
        self.norm = _Gain(dim)


class _ResidualFull(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.fn = _PreNormFull(dim)


class ConditionalUNet(_CondUNet):
    """denoising-sde ConditionalUNet(in_nc, out_nc, nf, depth=4); forward(x, time)."""

    def __init__(self, in_nc, out_nc, nf, depth=4):
        nn.Module.__init__(self)
        self.in_nc, self.out_nc, self.nf, self.depth = in_nc, out_nc, nf, depth
        time_dim = nf * 4
        self.init_conv = nn.Conv2d(in_nc, nf, 7, padding=3, bias=False)
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
        self.mid_attn = _ResidualFull(mid)
        self.mid_block2 = _ResBlock(mid, mid, time_dim)
        self.final_res_block = _ResBlock(nf * 2, nf, time_dim)
        self.final_conv = nn.Conv2d(nf, out_nc, 3, 1, 1)
        self._engine = None
        self._engine_key = None
        self.engine_flags = 0

    def _create_handle(self, L, device_index, flags):
        cfg = _lib.Config(self.in_nc, self.out_nc, self.nf, self.depth, device_index, flags | _lib.FLAG_UNCOND_FULLATTN)
        h = ctypes.c_void_p()
        _lib.check(L.irsde_create(ctypes.byref(cfg), ctypes.byref(h)))
        return h

    def forward(self, x, time):
        """noise = model(x, time) — denoising-sde/.../DenoisingUNet_arch.py:84-130."""
        ts = [int(time)] if isinstance(time, (int, float)) else [int(v) for v in torch.as_tensor(time).reshape(-1).tolist()]
        if x.device.type != "cuda":
            raise _lib.IrsdeError("ConditionalUNet.forward needs CUDA(HIP) tensors; got %s" % x.device)
        B, C, H, W = x.shape
        if len(ts) not in (1, B):
            raise _lib.IrsdeError("time must hold 1 or B timesteps")
        eng = self.engine(x.device)
        xin = x.detach().to(torch.float32).contiguous()
        out = torch.empty((B, self.out_nc, H, W), device=x.device, dtype=torch.float32)
        tarr = (ctypes.c_int64 * len(ts))(*ts)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().irsde_unet_forward(eng.h, ctypes.c_void_p(xin.data_ptr()), None, tarr, len(ts), B, H, W,
                                                     ctypes.c_void_p(out.data_ptr()), _lib.stream_ptr()))
        return out


class DenoisingSDE:
    def __init__(self, max_sigma, T, schedule="cosine", device=None):
        self.T = T
        self.device = device
        self.max_sigma = max_sigma / 255 if max_sigma > 1 else max_sigma  # sde_utils.py:379 ('>' here, '>=' in IRSDE)
        self.schedule = schedule
        self.seed = 0
        self.image_offset = 0
        self.injected_noise = None
        self.use_graph = True
        self._initialize(self.max_sigma, T, schedule)

    def _initialize(self, max_sigma, T, schedule, eps=0.04):  # sde_utils.py:382-426, same torch ops / order
        if schedule == "cosine":
            s = 0.008
            timesteps = T + 2
            x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float32)
            ac = torch.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
            ac = ac / ac[0]
            thetas = 1 - ac[1:-1]
        else:
            timesteps = T + 1
            scale = 1000 / timesteps
            thetas = torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float32)
        sigmas = torch.sqrt(max_sigma ** 2 * 2 * thetas)
        thetas_cumsum = torch.cumsum(thetas, dim=0) - thetas[0]
        self.dt = -1 / thetas_cumsum[-1] * math.log(eps)
        sigma_bars = torch.sqrt(max_sigma ** 2 * (1 - torch.exp(-2 * thetas_cumsum * self.dt)))
        self._cpu = dict(thetas=thetas, sigmas=sigmas, thetas_cumsum=thetas_cumsum, sigma_bars=sigma_bars)
        self.thetas = thetas.to(self.device)
        self.sigmas = sigmas.to(self.device)
        self.thetas_cumsum = thetas_cumsum.to(self.device)
        self.sigma_bars = sigma_bars.to(self.device)
        self.mu = 0.
        self.model = None
        tab = torch.zeros(T + 1, _lib.COEF_STRIDE, dtype=torch.float32)
        tab[:, 0] = thetas
        tab[:, 1] = sigmas
        tab[:, 2] = sigma_bars
        tab[:, 3] = self.dt
        tab[:, 4] = math.sqrt(self.dt)
        tab[:, 9] = torch.exp(-2 * thetas_cumsum * self.dt)
        self._coef = tab.contiguous()

    def set_model(self, model):
        self.model = model

    # ---- hot path: reverse_sde / reverse_ode (sde_utils.py:488-528) ---------------------------------
    def _run(self, ode, xt, x0, T, save_states, save_dir):
        T = self.T if int(T) < 0 else int(T)
        if not torch.is_tensor(xt) or xt.device.type != "cuda":
            raise _lib.IrsdeError("the DenoisingSDE sampler runs only on an AMD GPU through libirsde_hip.so (no CPU fallback)")
        x_in = xt.detach().to(torch.float32).contiguous()
        if T == 0:  # range(1, 1) is empty in the reference: the clone of xt comes back unchanged
            return x_in.clone()
        out = torch.empty_like(x_in)
        z = None
        if not ode and self.injected_noise is not None:
            z = self.injected_noise
            if z.device != xt.device or z.dtype != torch.float32 or tuple(z.shape[1:]) != tuple(xt.shape) or z.shape[0] < self.T + 1:
                raise _lib.IrsdeError("injected_noise must be float32 [T+1,B,C,H,W] on the sampling device")
            z = z.contiguous()
        B, C, H, W = x_in.shape
        L = _lib.lib()
        mode = _lib.MODE["dsde_ode" if ode else "dsde_sde"]
        m = _unwrap(self.model)
        interval = max(self.T // 100, 1)
        with torch.cuda.device(xt.device):
            stream = _lib.stream_ptr()
            # x0 replaces the model score only in reverse_sde (:489-493); reverse_ode always calls the model and uses x0
            # for nothing but the dumped [x, score, real_score] state image (:510-525)
            if (x0 is None or ode) and isinstance(m, ConditionalUNet) and not save_states:
                eng = m.engine(xt.device)
                key = ("dsde", self.T, self.schedule, self.max_sigma)
                if eng.schedule_key != key:
                    _lib.check(L.irsde_set_schedule(eng.h, self.T, ctypes.c_void_p(self._coef.data_ptr())))
                    eng.schedule_key = key
                flags = _lib.SAMPLE_GRAPH if self.use_graph else 0
                _lib.check(L.irsde_sample(eng.h, mode, ctypes.c_void_p(x_in.data_ptr()), None,
                                          ctypes.c_void_p(z.data_ptr()) if z is not None else None, self.seed,
                                          self.image_offset, B, H, W, T, 0, ctypes.c_void_p(out.data_ptr()), stream, flags))
                _check_fp16_range(self.model, out)
                return out
            out.copy_(x_in)
            for t in reversed(range(1, T + 1)):
                if x0 is not None and not ode:  # oracle score from the clean image (training / debugging aid, :492-493)
                    eps_hat = ((out - x0) / self.sigma_bars[t]).to(torch.float32).contiguous()
                else:
                    eps_hat = self.model(out, t).detach().to(torch.float32).contiguous()
                if save_states and ode and x0 is not None:
                    x_prev = out.clone()
                zt = ctypes.c_void_p(z[t].data_ptr()) if z is not None else None
                _lib.check(L.irsde_sde_step(mode, t, ctypes.c_void_p(self._coef[t].data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                            None, ctypes.c_void_p(eps_hat.data_ptr()), zt, self.seed, self.image_offset,
                                            B, C, H, W, stream))
                if save_states and t % interval == 0:
                    state = out
                    if ode and x0 is not None:  # :519-521 (score / real_score are those of the state BEFORE this step)
                        state = torch.cat([out, -eps_hat / self.sigma_bars[t], -(x_prev - x0) / self.sigma_bars[t] ** 2], dim=0)
                    _save_state(state, save_dir, t // interval)
        return out

    def reverse_sde(self, xt, x0=None, T=-1, save_states=False, save_dir="sde_state"):
        return self._run(False, xt, x0, T, save_states, save_dir)

    def reverse_ode(self, xt, x0=None, T=-1, save_states=False, save_dir="ode_state"):
        return self._run(True, xt, x0, T, save_states, save_dir)

    def get_optimal_timestep(self, sigma, eps=1e-6):
        """sde_utils.py:547-551."""
        sigma = sigma / 255 if sigma > 1 else sigma
        thetas_cumsum_hat = -1 / (2 * self.dt) * math.log(1 - sigma ** 2 / self.max_sigma ** 2 + eps)
        return torch.argmin((self.thetas_cumsum - thetas_cumsum_hat).abs())

    # ---- helper formulas of the reference API surface (training / analysis; plain tensor algebra) ---------
    def sigma(self, t):
        return self.sigmas[t]

    def theta(self, t):
        return self.thetas[t]

    def mu_bar(self, x0, t):
        return x0

    def sigma_bar(self, t):
        return self.sigma_bars[t]

    def drift(self, x, x0, t):
        return self.thetas[t] * (x0 - x) * self.dt

    def sde_reverse_drift(self, x, score, t):
        A = torch.exp(-2 * self.thetas_cumsum[t] * self.dt)
        return -0.5 * self.sigmas[t] ** 2 * (1 + A) * score * self.dt

    def ode_reverse_drift(self, x, score, t):
        A = torch.exp(-2 * self.thetas_cumsum[t] * self.dt)
        return -0.5 * self.sigmas[t] ** 2 * A * score * self.dt

    def dispersion(self, x, t):
        return self.sigmas[t] * (torch.randn_like(x) * math.sqrt(self.dt)).to(self.device)

    def get_score_from_noise(self, noise, t):
        return -noise / self.sigma_bar(t)

    def get_init_state_from_noise(self, x, noise, t):
        return x - self.sigma_bar(t) * noise

    def get_init_state_from_score(self, x, score, t):
        return x + self.sigma_bar(t) ** 2 * score

    def score_fn(self, x, t):
        return self.get_score_from_noise(self.model(x, t), t)

    def get_real_noise(self, xt, x0, t):
        return (xt - self.mu_bar(x0, t)) / self.sigma_bar(t)

    def get_real_score(self, xt, x0, t):
        return -(xt - self.mu_bar(x0, t)) / self.sigma_bar(t) ** 2

    def reverse_sde_step_mean(self, x, score, t):
        return x - self.sde_reverse_drift(x, score, t)

    def reverse_sde_step(self, x, score, t):
        return x - self.sde_reverse_drift(x, score, t) - self.dispersion(x, t)

    def reverse_ode_step(self, x, score, t):
        return x - self.ode_reverse_drift(x, score, t)

    def reverse_optimum_step(self, xt, x0, t):
        A = torch.exp(-self.thetas[t] * self.dt)
        B = torch.exp(-self.thetas_cumsum[t] * self.dt)
        C = torch.exp(-self.thetas_cumsum[t - 1] * self.dt)
        term1 = A * (1 - C ** 2) / (1 - B ** 2)
        term2 = C * (1 - A ** 2) / (1 - B ** 2)
        return term1 * (xt - x0) + term2 * (x0 - x0) + x0

    def optimal_reverse(self, xt, x0, T=-1):
        T = self.T if T < 0 else T
        x = xt.clone()
        for t in reversed(range(1, T + 1)):
            x = self.reverse_optimum_step(x, x0, t)
        return x

    def weights(self, t):
        return self.sigmas[t] ** 2

    def generate_random_states(self, x0):
        x0 = x0.to(self.device)
        timesteps = torch.randint(1, self.T + 1, (x0.shape[0], 1, 1, 1)).long()
        noises = torch.randn_like(x0, dtype=torch.float32)
        return timesteps, noises * self.sigma_bar(timesteps) + x0
