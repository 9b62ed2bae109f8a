"""`IRSDE` — the reference's SDE object (codes/utils/sde_utils.py:80-361) on the HIP engine.

Same constructor, attributes and method names as the reference class, so
`sde = IRSDE(max_sigma, T, schedule, eps, device); sde.set_model(model.model); model.test(sde, ...)`
(deraining/test.py:70-72,104-108) works unchanged.

Hot path: `reverse_sde / reverse_ode / reverse_posterior`.
  * model is this package's `ConditionalUNet` (possibly DataParallel/DDP-wrapped): the whole T-step loop
    runs inside libirsde_hip.so (`irsde_sample`: hand-written HIP kernels, one hipGraph replayed per step).
  * any other callable score model (e.g. a NAFNet nn.Module, or kwargs such as bokeh `lens_info`): the
    model is called per step as the reference does and the state update runs in the fused HIP kernel
    `irsde_sde_step`.
There is no PyTorch/CPU fallback for the sampler: CPU tensors raise.

Schedule tables are built with the same torch CPU ops in the same order as the reference
(sde_utils.py:84-152) so they are bit-identical to its tables (tests/test_host_logic.py pins this against
tests/golden/schedule.npz).

RNG: the reference draws `torch.randn_like` per step (sde_utils.py:182,223).  Here: set
`sde.injected_noise` ([T+1,B,C,H,W], index t) for parity runs, otherwise Philox4x32-10 keyed by
(`sde.seed`, global image index, t, element) — independent of how a batch is sharded over GPUs.

The remaining methods (`mu_bar`, `generate_random_states`, `reverse_optimum_step`, ...) are the
training-time helper formulas of the reference API surface (train.py:238, denoising_model.py:135-140).
Training is outside this repo's scope; they are provided as the same few lines of tensor
algebra so that callers of the full `IRSDE` surface keep working.
"""
import ctypes
import math
import os

import torch

from . import _lib
from .unet import ConditionalUNet


def _check_fp16_range(model, out):
    """The fp16-operand modes ('fp32_split_f16', 'fp16') are range-limited (|activation| below ~1e4, include/irsde_hip.h): leaving that
    window shows up as inf / NaN in the sampler's output.  Fail loudly instead of handing it on (costs one stream synchronisation, in
    these opt-in modes only; `model.check_fp16_range = False` switches it off)."""
    m = _unwrap(model)
    flags = int(getattr(m, "engine_flags", 0))
    if flags & (_lib.FLAG_SPLIT_F16X2 | _lib.FLAG_FP16) and getattr(m, "check_fp16_range", True):
        if not bool(torch.isfinite(out).all()):
            raise _lib.IrsdeError("non-finite sampler output in an fp16-operand mode: the activations left fp16's range "
                                  "(|x| < ~1e4 is required) — use compute dtype 'fp32' or 'fp32_split' (bf16 pieces, f32's range)")


def _unwrap(model):
    m = model
    while hasattr(m, "module") and isinstance(getattr(m, "module"), torch.nn.Module):
        m = m.module
    return m


class IRSDE:
    def __init__(self, max_sigma, T=100, schedule="cosine", eps=0.01, device=None):
        self.T = T
        self.device = device
        self.max_sigma = max_sigma / 255 if max_sigma >= 1 else max_sigma
        self.schedule = schedule
        self.eps = eps
        self.seed = 0
        self.image_offset = 0          # global index of this shard's first image (multi-GPU sharding)
        self.injected_noise = None     # [T+1,B,C,H,W] on device => parity mode
        self.use_graph = True
        self.profile = False
        self._initialize(self.max_sigma, T, schedule, eps)

    # ---- schedule (sde_utils.py:88-152), same torch CPU ops / order as the reference ----------
    def _initialize(self, max_sigma, T, schedule, eps=0.01):
        if schedule == "cosine":
            s = 0.008
            timesteps = T + 2
            x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float32)
            ac = torch.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
            ac = ac / ac[0]
            thetas = 1 - ac[1:-1]
        elif schedule == "linear":
            timesteps = T + 1
            scale = 1000 / timesteps
            thetas = torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float32)
        elif schedule == "constant":
            thetas = torch.ones(T + 1, dtype=torch.float32)
        else:
            raise ValueError("Not implemented such schedule yet!!! (%r)" % (schedule,))
        sigmas = torch.sqrt(max_sigma ** 2 * 2 * thetas)
        thetas_cumsum = torch.cumsum(thetas, dim=0) - thetas[0]
        self.dt = -1 / thetas_cumsum[-1] * math.log(eps)  # 0-dim fp32 CPU tensor, as in the reference (:143)
        sigma_bars = torch.sqrt(max_sigma ** 2 * (1 - torch.exp(-2 * thetas_cumsum * self.dt)))
        self._cpu = dict(thetas=thetas, sigmas=sigmas, thetas_cumsum=thetas_cumsum, sigma_bars=sigma_bars)
        self.thetas = thetas.to(self.device)
        self.sigmas = sigmas.to(self.device)
        self.thetas_cumsum = thetas_cumsum.to(self.device)
        self.sigma_bars = sigma_bars.to(self.device)
        self.mu = 0.
        self.model = None
        self._coef = self._coef_table()

    def _coef_table(self):
        """Per-step coefficient rows for the HIP update kernel (layout: include/irsde_hip.h), computed with
        the reference's own fp32 expressions (sde_utils.py:197-217, 237-239) on the CPU tables."""
        c = self._cpu
        T, dt = self.T, self.dt
        th, cs = c["thetas"], c["thetas_cumsum"]
        tab = torch.zeros(T + 1, _lib.COEF_STRIDE, dtype=torch.float32)
        tab[:, 0] = th
        tab[:, 1] = c["sigmas"]
        tab[:, 2] = c["sigma_bars"]
        tab[:, 3] = dt
        tab[:, 4] = math.sqrt(dt)
        for t in range(1, T + 1):
            A = torch.exp(-th[t] * dt)
            B = torch.exp(-cs[t] * dt)
            C = torch.exp(-cs[t - 1] * dt)
            tab[t, 5] = torch.exp(cs[t] * dt)
            tab[t, 6] = A * (1 - C ** 2) / (1 - B ** 2)
            tab[t, 7] = C * (1 - A ** 2) / (1 - B ** 2)
            A2 = torch.exp(-2 * th[t] * dt)
            B2 = torch.exp(-2 * cs[t] * dt)
            C2 = torch.exp(-2 * cs[t - 1] * dt)
            var = (1 - A2) * (1 - C2) / (1 - B2)
            logv = torch.log(torch.clamp(var, min=(1e-20 * dt)))
            tab[t, 8] = (0.5 * logv).exp() * self.max_sigma
        return tab.contiguous()

    # ---- reference setters --------------------------------------------------------------------
    def set_mu(self, mu):
        self.mu = mu

    def set_model(self, model):
        self.model = model

    # ---- the hot path ---------------------------------------------------------------------------
    def _engine_for(self, x):
        m = _unwrap(self.model)
        if not isinstance(m, ConditionalUNet):
            return None
        eng = m.engine(x.device)
        key = (self.T, self.schedule, self.eps, self.max_sigma)
        if eng.schedule_key != key:
            with torch.cuda.device(x.device):
                _lib.check(_lib.lib().irsde_set_schedule(eng.h, self.T, ctypes.c_void_p(self._coef.data_ptr())))
            eng.schedule_key = key
        return eng

    def _check_inputs(self, xt):
        if not torch.is_tensor(xt) or xt.device.type != "cuda":
            raise _lib.IrsdeError("the IR-SDE sampler runs only on an AMD GPU through libirsde_hip.so; "
                                  "got a %s tensor (no CPU fallback)" % getattr(xt, "device", type(xt)))
        if not torch.is_tensor(self.mu) or self.mu.shape != xt.shape:
            raise _lib.IrsdeError("set_mu(LQ) with the same shape as the state must be called before sampling")

    def _noise_ptr(self, xt, need):
        if not need or self.injected_noise is None:
            return None, None
        z = self.injected_noise
        if z.device != xt.device or z.dtype != torch.float32 or tuple(z.shape[1:]) != tuple(xt.shape) \
                or z.shape[0] < self.T + 1:
            raise _lib.IrsdeError("injected_noise must be float32 [T+1,B,C,H,W] on the sampling device")
        return z.contiguous(), None

    def _run(self, mode, xt, T, save_states, save_dir, kwargs):
        T = self.T if T < 0 else T
        self._check_inputs(xt)
        x_in = xt.detach().to(torch.float32).contiguous()
        if int(T) == 0:  # range(1, 1) is empty in the reference: the clone of xt comes back unchanged
            return x_in.clone()
        mu = self.mu.detach().to(device=xt.device, dtype=torch.float32).contiguous()
        out = torch.empty_like(x_in)
        z, _ = self._noise_ptr(x_in, mode != "ode")
        zp = ctypes.c_void_p(z.data_ptr()) if z is not None else None
        B, C, H, W = x_in.shape
        if set(kwargs) == {"lens_info"} and hasattr(_unwrap(self.model), "set_lens_info"):
            # latent-bokeh: the lens FiLM is constant over the loop -> evaluate it once and stay on the engine path
            _unwrap(self.model).set_lens_info(kwargs["lens_info"], B, x_in.device)
            kwargs = {}
        eng = None if kwargs else self._engine_for(x_in)
        L = _lib.lib()
        interval = max(self.T // 100, 1)
        with torch.cuda.device(xt.device):
            stream = _lib.stream_ptr()
            if eng is not None:
                flags = (_lib.SAMPLE_PROFILE if self.profile else 0) | (_lib.SAMPLE_GRAPH if self.use_graph else 0)
                if not save_states:
                    _lib.check(L.irsde_sample(eng.h, _lib.MODE[mode], ctypes.c_void_p(x_in.data_ptr()),
                                              ctypes.c_void_p(mu.data_ptr()), zp, self.seed, self.image_offset,
                                              B, H, W, T, 0, ctypes.c_void_p(out.data_ptr()), stream, flags))
                else:
                    # dump every `interval` steps like the reference (:260-264): run the loop in segments
                    cur, t = x_in, T
                    while t > 0:
                        d = (t // interval) * interval      # next step after which the reference dumps
                        t_stop = d - 1 if d >= 1 else 0
                        _lib.check(L.irsde_sample(eng.h, _lib.MODE[mode], ctypes.c_void_p(cur.data_ptr()),
                                                  ctypes.c_void_p(mu.data_ptr()), zp, self.seed, self.image_offset,
                                                  B, H, W, t, t_stop, ctypes.c_void_p(out.data_ptr()), stream, flags))
                        if d >= 1:
                            _save_state(out, save_dir, d // interval)
                        cur, t = out, t_stop
                _check_fp16_range(self.model, out)
                return out
            # foreign score model: reference-style loop, fused HIP state update per step
            out.copy_(x_in)
            coef = self._coef
            for t in reversed(range(1, T + 1)):
                eps_hat = self.model(out, self.mu, t, **kwargs).detach().to(torch.float32).contiguous()
                zt = ctypes.c_void_p(z[t].data_ptr()) if z is not None else None
                _lib.check(L.irsde_sde_step(_lib.MODE[mode], t, ctypes.c_void_p(coef[t].data_ptr()),
                                            ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(mu.data_ptr()),
                                            ctypes.c_void_p(eps_hat.data_ptr()), zt, self.seed, self.image_offset,
                                            B, C, H, W, stream))
                if save_states and t % interval == 0:
                    _save_state(out, save_dir, t // interval)
        return out

    def reverse_sde(self, xt, T=-1, save_states=False, save_dir="sde_state", **kwargs):
        """sde_utils.py:252-266."""
        return self._run("sde", xt, T, save_states, save_dir, kwargs)

    def reverse_ode(self, xt, T=-1, save_states=False, save_dir="ode_state", **kwargs):
        """sde_utils.py:268-282."""
        return self._run("ode", xt, T, save_states, save_dir, kwargs)

    def reverse_posterior(self, xt, T=-1, save_states=False, save_dir="posterior_state", **kwargs):
        """sde_utils.py:284-299."""
        return self._run("posterior", xt, T, save_states, save_dir, kwargs)

    def noise_state(self, tensor):
        """x_T = LQ + N(0,1) * max_sigma (sde_utils.py:360-361; called on a CPU tensor by test.py:104)."""
        return tensor + torch.randn_like(tensor) * self.max_sigma

    def last_profile(self):
        """Per-kernel-class timing of the last `profile=True` sampling call (irsde_get_profile)."""
        m = _unwrap(self.model)
        out = (ctypes.c_double * 12)()
        _lib.check(_lib.lib().irsde_get_profile(m.engine().h, out))
        keys = ["conv_ms", "conv_flops", "conv_launches", "conv_bytes", "ln_ms", "attn_ms", "other_ms", "wall_ms",
                "net_evals", "wino_ms", "conv_exec_flops", "reserved"]
        return dict(zip(keys, list(out)))

    # ---- model plumbing (sde_utils.py:184-194) ------------------------------------------------
    def sigma_bar(self, t):
        return self.sigma_bars[t]

    def sigma(self, t):
        return self.sigmas[t]

    def theta(self, t):
        return self.thetas[t]

    def get_score_from_noise(self, noise, t):
        return -noise / self.sigma_bar(t)

    def score_fn(self, x, t, **kwargs):
        return self.get_score_from_noise(self.model(x, self.mu, t, **kwargs), t)

    def noise_fn(self, x, t, **kwargs):
        return self.model(x, self.mu, t, **kwargs)

    # ---- training-time helper formulas of the reference API surface (see module docstring) -----
    def mu_bar(self, x0, t):
        return self.mu + (x0 - self.mu) * torch.exp(-self.thetas_cumsum[t] * self.dt)

    def drift(self, x, t):
        return self.thetas[t] * (self.mu - x) * self.dt

    def dispersion(self, x, t):
        return self.sigmas[t] * (torch.randn_like(x) * math.sqrt(self.dt)).to(self.device)

    def sde_reverse_drift(self, x, score, t):
        return (self.thetas[t] * (self.mu - x) - self.sigmas[t] ** 2 * score) * self.dt

    def ode_reverse_drift(self, x, score, t):
        return (self.thetas[t] * (self.mu - x) - 0.5 * self.sigmas[t] ** 2 * score) * self.dt

    def forward_step(self, x, t):
        return x + self.drift(x, t) + self.dispersion(x, t)

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
        return term1 * (xt - self.mu) + term2 * (x0 - self.mu) + self.mu

    def reverse_optimum_std(self, t):
        A = torch.exp(-2 * self.thetas[t] * self.dt)
        B = torch.exp(-2 * self.thetas_cumsum[t] * self.dt)
        C = torch.exp(-2 * self.thetas_cumsum[t - 1] * self.dt)
        posterior_var = (1 - A) * (1 - C) / (1 - B)
        min_value = (1e-20 * self.dt).to(self.device)
        return (0.5 * torch.log(torch.clamp(posterior_var, min=min_value))).exp() * self.max_sigma

    def get_init_state_from_noise(self, xt, noise, t):
        return (xt - self.mu - self.sigma_bar(t) * noise) * torch.exp(self.thetas_cumsum[t] * self.dt) + self.mu

    def reverse_posterior_step(self, xt, noise, t):
        x0 = self.get_init_state_from_noise(xt, noise, t)
        return self.reverse_optimum_step(xt, x0, t) + self.reverse_optimum_std(t) * torch.randn_like(xt)

    def get_real_noise(self, xt, x0, t):
        return (xt - self.mu_bar(x0, t)) / self.sigma_bar(t)

    def get_real_score(self, xt, x0, t):
        return -(xt - self.mu_bar(x0, t)) / self.sigma_bar(t) ** 2

    def forward(self, x0, T=-1, save_dir="forward_state"):
        T = self.T if T < 0 else T
        x = x0.clone()
        for t in range(1, T + 1):
            x = self.forward_step(x, t)
            _save_state(x, save_dir, t, prefix="state_")
        return x

    def optimal_reverse(self, xt, x0, T=-1):
        T = self.T if T < 0 else T
        x = xt.clone()
        for t in reversed(range(1, T + 1)):
            x = self.reverse_optimum_step(x, x0, t)
        return x

    def weights(self, t):
        return torch.exp(-self.thetas_cumsum[t] * self.dt)

    def generate_random_states(self, x0, mu):
        x0 = x0.to(self.device)
        mu = mu.to(self.device)
        self.set_mu(mu)
        batch = x0.shape[0]
        timesteps = torch.randint(1, self.T + 1, (batch, 1, 1, 1)).long()
        state_mean = self.mu_bar(x0, timesteps)
        noises = torch.randn_like(state_mean)
        noise_level = self.sigma_bar(timesteps)
        noisy_states = noises * noise_level + state_mean
        return timesteps, noisy_states.to(torch.float32)


def _save_state(x, save_dir, idx, prefix="state_"):
    """PNG dump of a state batch (the reference uses torchvision.utils.save_image, sde_utils.py:264)."""
    os.makedirs(save_dir, exist_ok=True)
    try:
        from PIL import Image
    except ImportError as ex:  # pragma: no cover
        raise _lib.IrsdeError("save_states=True needs Pillow") from ex
    img = x.detach().float().clamp(0, 1).cpu()
    b, c, h, w = img.shape
    ncol = min(8, b)
    nrow = (b + ncol - 1) // ncol
    pad = 2
    grid = torch.zeros(c, nrow * (h + pad) + pad, ncol * (w + pad) + pad)
    for i in range(b):
        r, q = divmod(i, ncol)
        grid[:, pad + r * (h + pad): pad + r * (h + pad) + h, pad + q * (w + pad): pad + q * (w + pad) + w] = img[i]
    arr = (grid * 255 + 0.5).clamp(0, 255).to(torch.uint8).permute(1, 2, 0).numpy()
    if arr.shape[2] == 1:
        arr = arr[:, :, 0]
    Image.fromarray(arr).save(os.path.join(save_dir, "%s%d.png" % (prefix, idx)))
