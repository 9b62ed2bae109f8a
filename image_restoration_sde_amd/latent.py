"""Latent-Refusion wrapper: the frozen latent compressor `UNet`, the latent-task
`ConditionalNAFNet`, and the inference half of `latent_denoising_model.DenoisingModel`.

Reference: /root/reference/codes/config/latent-dehazing/
    models/modules/UNet_arch.py:17-96            UNet(in_ch, out_ch, ch, ch_mult, embed_dim): encode / decode / forward
    models/modules/DenoisingNAFNet_arch.py:85-186 ConditionalNAFNet whose `ending` sees x + intro(x)
    models/latent_denoising_model.py:40-51,146-152,177-200
    test.py:90-100     latent_LQ, hidden = model.encode(LQ); noisy = sde.noise_state(latent_LQ);
                       model.feed_data(noisy, latent_LQ); model.test(sde, hidden)
(the latent-bokeh variant additionally threads `lens_info` through every NAFBlock; not built.)

Modules only own parameters under the reference's state_dict names; the arithmetic runs in libirsde_hip.so.
"""
import ctypes
import weakref
from collections import OrderedDict
from collections.abc import Sequence

import torch
import torch.nn as nn

from . import _lib
from .nafnet import ConditionalNAFNet as _ImageNAFNet
from .unet import _Block, _Engine, _Residual


class _PlainResBlock(nn.Module):  # module_util.ResBlock with time_emb_dim=None (latent-dehazing module_util.py:132-153)
    def __init__(self, ci, co):
        super().__init__()
        self.block1 = _Block(ci, co)
        self.block2 = _Block(co, co)
        self.res_conv = nn.Conv2d(ci, co, 1, bias=False) if ci != co else nn.Identity()


class _ResidentHidden(Sequence):
    """The `hidden` list of UNet.encode (UNet_arch.py:59-77) while its tensors still sit in the engine's working layout.  The reference only
    hands the list on to decode; touching it (indexing, iterating, list(...)) materialises the NCHW tensors first, and so do a second encode or a
    decode with other skips on the same model — the object never dangles.  (A Sequence, not a list subclass: C-level list fast paths would read
    the empty base storage of an unmaterialised subclass.)  Differences from the reference's plain list: `isinstance(h, list)` is False and
    there is no `append`; `tolist()` (also what pickle / `torch.save` / `h + [...]` produce) gives the plain list, and `UNet.resident_hidden =
    False` makes `encode` return one right away.  After `materialize()` the object drops its engine reference."""

    def __init__(self, eng, B, H, W, shapes, device):
        self._eng, self._geom, self._shapes, self._device = eng, (B, H, W), list(shapes), device
        self._items = None

    @property
    def _mat(self):
        return self._items is not None

    def in_place(self, eng, B, H, W):
        return self._items is None and eng is self._eng and self._geom == (B, H, W)

    def materialize(self):
        if self._items is None:
            B, H, W = self._geom
            items = []
            with torch.cuda.device(self._device):
                for k, shp in enumerate(self._shapes):
                    t = torch.empty((B,) + tuple(shp), device=self._device, dtype=torch.float32)
                    _lib.check(_lib.lib().irsde_latent_hidden(self._eng.h, B, H, W, k, ctypes.c_void_p(t.data_ptr()), _lib.stream_ptr()))
                    items.append(t)
            self._items = items
            self._eng = None   # the tensors own the data now: do not pin the engine (its weights, arena and plans) any longer
        return self._items

    def tolist(self):
        """The reference's plain `list` of NCHW tensors (for `h + [...]`, `h.append`, `torch.save`: anything beyond Sequence)."""
        return list(self.materialize())

    def __reduce__(self):
        # pickle / copy / torch.save see the reference's type: a plain list of tensors (a ctypes engine handle cannot travel)
        return (list, (self.tolist(),))

    def __add__(self, other):
        return self.tolist() + list(other)

    def __radd__(self, other):
        return list(other) + self.tolist()

    def __len__(self):
        return len(self._shapes)

    def __getitem__(self, i):
        return self.materialize()[i]

    def __iter__(self):
        return iter(self.materialize())

    def __repr__(self):
        return "<hidden states of UNet.encode: %d skips, %s>" % (len(self._shapes), "NCHW tensors" if self._mat else "resident in the engine")


class UNet(nn.Module):
    def __init__(self, in_ch=3, out_ch=3, ch=64, ch_mult=[1, 2, 4, 4], embed_dim=4):
        super().__init__()
        self.in_ch, self.out_ch, self.ch, self.ch_mult, self.embed_dim = in_ch, out_ch, ch, list(ch_mult), embed_dim
        self.depth = len(ch_mult)
        self.init_conv = nn.Conv2d(in_ch, ch, 3, padding=1, bias=False)
        self.encoder, self.decoder = nn.ModuleList([]), nn.ModuleList([])
        mult = [1] + list(ch_mult)
        for i in range(self.depth):
            di, do = ch * mult[i], ch * mult[i + 1]
            last = i == self.depth - 1
            self.encoder.append(nn.ModuleList([
                _PlainResBlock(di, di), _PlainResBlock(di, di), _Residual(di) if last else nn.Identity(),
                nn.Conv2d(di, do, 4, 2, 1) if not last else nn.Conv2d(di, do, 3, padding=1, bias=False)]))
            self.decoder.insert(0, nn.ModuleList([
                _PlainResBlock(do + di, do), _PlainResBlock(do + di, do), _Residual(do) if last else nn.Identity(),
                nn.Sequential(nn.Identity(), nn.Conv2d(do, di, 3, 1, 1)) if i != 0 else nn.Conv2d(do, di, 3, padding=1, bias=False)]))
        mid = ch * mult[-1]
        self.latent_conv = nn.Conv2d(mid, embed_dim, 1, bias=False)
        self.post_latent_conv = nn.Conv2d(embed_dim, mid, 1, bias=False)
        self.final_conv = nn.Conv2d(ch, out_ch, 3, 1, 1)
        self._engine = None
        self._engine_key = None
        self.engine_flags = 0
        self.H = self.W = None
        self._live_hidden = None   # weakref to the lazy hidden list of the last encode (resident skips)

    # ---- engine management (as ConditionalUNet) ----------------------------------------------
    def _create_handle(self, L, device_index, flags):
        cfg = _lib.LatentConfig()
        cfg.in_ch, cfg.out_ch, cfg.ch, cfg.n_mult, cfg.embed_dim = self.in_ch, self.out_ch, self.ch, self.depth, self.embed_dim
        for i, v in enumerate(self.ch_mult):
            cfg.ch_mult[i] = v
        cfg.device, cfg.flags = device_index, flags
        h = ctypes.c_void_p()
        _lib.check(L.irsde_create_latent_unet(ctypes.byref(cfg), ctypes.byref(h)))
        return h

    def engine(self, device=None):
        if device is None:
            device = next(self.parameters()).device
        if device.type != "cuda":
            raise _lib.IrsdeError("the latent UNet runs only on an AMD GPU through libirsde_hip.so (no CPU/PyTorch fallback)")
        key = (device.index if device.index is not None else torch.cuda.current_device(), self.engine_flags,
               tuple((p.data_ptr(), p._version) for p in self.parameters()))
        if self._engine is None or self._engine_key != key:
            old = self._live_hidden() if self._live_hidden is not None else None
            if old is not None:
                old.materialize()   # (the old engine, which the list keeps alive, still holds its skips)
            self._engine = _Engine(self, key[0], self.engine_flags)
            self._engine_key = key
        return self._engine

    def _shapes(self, eng, H, W):
        L = _lib.lib()
        lat = (ctypes.c_int64 * 3)()
        n = ctypes.c_int()
        hid = (ctypes.c_int64 * (3 * (2 * self.depth + 1)))()
        _lib.check(L.irsde_latent_shapes(eng.h, H, W, lat, hid, ctypes.byref(n)))
        return tuple(lat), [tuple(hid[3 * k:3 * k + 3]) for k in range(n.value)]

    # ---- reference interface (UNet_arch.py:59-96) ---------------------------------------------
    resident_hidden = True   # encode() leaves the skips in the engine and returns a lazy list (ABI 104); False: NCHW tensors right away

    def encode(self, x):
        if x.device.type != "cuda":
            raise _lib.IrsdeError("UNet.encode needs CUDA(HIP) tensors; got %s" % x.device)
        self.H, self.W = x.shape[2:]
        B = x.shape[0]
        eng = self.engine(x.device)
        lat_s, hid_s = self._shapes(eng, self.H, self.W)
        xin = x.detach().to(torch.float32).contiguous()
        latent = torch.empty((B,) + lat_s, device=x.device, dtype=torch.float32)
        # a lazy list of an earlier encode still points at the engine's skip storage: give it its tensors before they are overwritten
        old = self._live_hidden() if self._live_hidden is not None else None
        if old is not None:
            old.materialize()
        if self.resident_hidden:
            with torch.cuda.device(x.device):
                _lib.check(_lib.lib().irsde_latent_encode(eng.h, ctypes.c_void_p(xin.data_ptr()), B, self.H, self.W,
                                                          ctypes.c_void_p(latent.data_ptr()), None, _lib.stream_ptr()))
            hidden = _ResidentHidden(eng, B, self.H, self.W, hid_s, x.device)
            self._live_hidden = weakref.ref(hidden)
            return latent, hidden
        self._live_hidden = None
        hidden = [torch.empty((B,) + s, device=x.device, dtype=torch.float32) for s in hid_s]
        ptrs = (ctypes.c_void_p * len(hidden))(*[h.data_ptr() for h in hidden])
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().irsde_latent_encode(eng.h, ctypes.c_void_p(xin.data_ptr()), B, self.H, self.W,
                                                      ctypes.c_void_p(latent.data_ptr()), ptrs, _lib.stream_ptr()))
        return latent, hidden

    def decode(self, x, h):
        if self.H is None:
            raise _lib.IrsdeError("UNet.decode before encode: the crop size comes from the encoded image (UNet_arch.py:60,91)")
        if x.device.type != "cuda":
            raise _lib.IrsdeError("UNet.decode needs CUDA(HIP) tensors; got %s" % x.device)
        B = x.shape[0]
        eng = self.engine(x.device)
        lat_s, hid_s = self._shapes(eng, self.H, self.W)
        if isinstance(h, _ResidentHidden) and h.in_place(eng, B, self.H, self.W):
            # the reference only threads `hidden` from encode to decode (latent_denoising_model.py:177-189): the skips never left the engine
            if tuple(x.shape[1:]) != lat_s:
                raise _lib.IrsdeError("latent shape does not belong to the encoded %dx%d image" % (self.H, self.W))
            lat = x.detach().to(torch.float32).contiguous()
            out = torch.empty((B, self.out_ch, self.H, self.W), device=x.device, dtype=torch.float32)
            with torch.cuda.device(x.device):
                _lib.check(_lib.lib().irsde_latent_decode(eng.h, ctypes.c_void_p(lat.data_ptr()), None, B, self.H, self.W,
                                                          ctypes.c_void_p(out.data_ptr()), _lib.stream_ptr()))
            return out
        if tuple(x.shape[1:]) != lat_s or len(h) != len(hid_s) or any(tuple(t.shape[1:]) != s for t, s in zip(h, hid_s)):
            raise _lib.IrsdeError("latent / hidden shapes do not belong to the encoded %dx%d image" % (self.H, self.W))
        live = self._live_hidden() if self._live_hidden is not None else None
        if live is not None and live is not h:
            live.materialize()   # this decode overwrites the engine's skip storage with the caller's tensors
        lat = x.detach().to(torch.float32).contiguous()
        hid = [t.detach().to(torch.float32).contiguous() for t in h]
        ptrs = (ctypes.c_void_p * len(hid))(*[t.data_ptr() for t in hid])
        out = torch.empty((B, self.out_ch, self.H, self.W), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().irsde_latent_decode(eng.h, ctypes.c_void_p(lat.data_ptr()), ptrs, B, self.H, self.W,
                                                      ctypes.c_void_p(out.data_ptr()), _lib.stream_ptr()))
        return out

    def forward(self, x):
        x, h = self.encode(x)
        return self.decode(x, h)


class ConditionalNAFNet(_ImageNAFNet):
    """latent-dehazing ConditionalNAFNet: same parameters as the image-space one (668 tensors for nasde.yml), but
    `ending(x + intro(x))` (DenoisingNAFNet_arch.py:162-176) and typically img_channel = the latent's embed_dim (<= 8)."""

    def _create_handle(self, L, device_index, flags):
        return super()._create_handle(L, device_index, flags | _lib.FLAG_NAF_INTRO_SKIP)


def define_G(opt):
    """latent-*/models/networks.py: network_G.which_model looked up by name in the task's own modules package — the
    latent-bokeh task (`distortion: bokeh`, options/bokeh/test/refusion.yml:4) has the lens-conditioned ConditionalNAFNet."""
    o = opt["network_G"]
    name = o.get("which_model", o.get("which_model_G"))
    if name != "ConditionalNAFNet":
        raise NotImplementedError("latent score network [%s] is not built (ConditionalNAFNet only)" % name)
    if opt.get("distortion") == "bokeh":
        from .latent_bokeh import ConditionalNAFNet as BokehNAFNet
        return BokehNAFNet(**o["setting"])
    return ConditionalNAFNet(**o["setting"])


def define_L(opt):
    o = opt["network_L"]
    if o["which_model"] != "UNet":
        raise NotImplementedError("latent model [%s] not recognized" % o["which_model"])
    return UNet(**o["setting"])


class LatentDenoisingModel:
    """Inference surface of latent_denoising_model.DenoisingModel (:40-51 construction, :146-152 feed_data,
    :177-200 test / get_current_visuals, :222-231 load)."""

    def __init__(self, opt):
        self.opt = opt
        self.device = torch.device("cuda")
        if opt.get("is_train", False):
            raise NotImplementedError("training is out of scope of the MI355X sampler (SURVEY.md §2)")
        self.model = define_G(opt).to(self.device)
        self.latent_model = define_L(opt).to(self.device)
        for p in self.latent_model.parameters():
            p.requires_grad = False
        self.load()
        self.encode = self.latent_model.encode
        self.decode = self.latent_model.decode

    def feed_data(self, state, LQ, GT=None, src_lens=None, tgt_lens=None, disparity=None, alpha=None):
        """latent-dehazing :146-152; latent-bokeh adds the lens triple (:143-153)."""
        self.state = state.to(self.device)
        self.condition = LQ.to(self.device)
        self.state_0 = GT.to(self.device) if GT is not None else None
        self.src_lens, self.tgt_lens, self.disparity, self.alpha = src_lens, tgt_lens, disparity, alpha

    def test(self, sde=None, hidden=None, perform_ode=False, save_states=False):
        sde.set_mu(self.condition)
        kw = {}
        if getattr(self, "src_lens", None) is not None:  # latent-bokeh :183-189 (there only reverse_sde gets the kwargs; the
            kw["lens_info"] = [self.src_lens, self.tgt_lens, self.disparity]  # ODE branch would fail without them)
        self.model.eval()
        with torch.no_grad():
            if not perform_ode:
                latent = sde.reverse_sde(self.state, save_states=save_states, **kw)
            else:
                latent = sde.reverse_ode(self.state, save_states=save_states, **kw)
            self.output = self.decode(latent, hidden)
        self.model.train()

    def get_current_visuals(self, need_GT=True):
        out = OrderedDict()
        out["Input"] = self.condition.detach()[0].float().cpu()
        out["Output"] = self.output.detach()[0].float().cpu()
        if need_GT and self.state_0 is not None:
            out["GT"] = self.state_0.detach()[0].float().cpu()
        return out

    def load(self):
        path = self.opt.get("path") or {}
        strict = path.get("strict_load", True)
        if path.get("pretrain_model_G") is not None:
            self.load_network(path["pretrain_model_G"], self.model, strict)
        if path.get("pretrain_model_L") is not None:
            self.load_network(path["pretrain_model_L"], self.latent_model, strict)

    @staticmethod
    def load_network(load_path, network, strict=True):
        load_net = torch.load(load_path, map_location="cpu")
        clean = OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in load_net.items())
        network.load_state_dict(clean, strict=strict)
