"""latent-bokeh score network: `ConditionalNAFNet` conditioned on lens information.

Reference: /root/reference/codes/config/latent-bokeh/models/modules/DenoisingNAFNet_arch.py
    :15-91   NAFBlock with `time_mlp` (4c) and `cam_mlp` (2c): FiLM `x * (cam_scale + 1) + cam_shift` on the gated FFN activation
    :94-158  ConditionalNAFNet: SinusoidalPosEmb outside `time_mlp` (indices 0 / 2), `cam_mlp` over the three lens embeddings
    :159-198 forward(inp, cond, time, **kwargs) with kwargs['lens_info'] = [src_lens, tgt_lens, disparity]
and latent_denoising_model.py:183-189 (`sde.reverse_sde(self.state, lens_info=lens_info)`).

Same HIP engine as the other NAFNets (IRSDE_FLAG_NAF_LENS): the lens FiLM rows are evaluated once per call
(`irsde_set_lens_info`), the per-block FiLM runs inside the conv4 SimpleGate epilogue.  Unlike the reference (whose
int-time path only works for one image) every image of a batch carries its own lens triple.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib
from .nafnet import ConditionalNAFNet as _ImageNAFNet
from .unet import _Gain


class _LensNAFBlock(nn.Module):  # parameter container, reference names
    def __init__(self, c, time_emb_dim):
        super().__init__()
        self.time_mlp = nn.Sequential(nn.Identity(), nn.Linear(time_emb_dim // 2, c * 4))
        self.cam_mlp = nn.Sequential(nn.Identity(), nn.Linear(time_emb_dim // 2, c * 2))
        self.conv1 = nn.Conv2d(c, 2 * c, 1)
        self.conv2 = nn.Conv2d(2 * c, 2 * c, 3, padding=1, groups=2 * c)
        self.conv3 = nn.Conv2d(c, c, 1)
        self.sca = nn.Sequential(nn.Identity(), nn.Conv2d(c, c, 1))
        self.conv4 = nn.Conv2d(c, 2 * c, 1)
        self.conv5 = nn.Conv2d(c, c, 1)
        self.norm1 = _Gain(c)
        self.norm2 = _Gain(c)
        self.beta = nn.Parameter(torch.zeros((1, c, 1, 1)))
        self.gamma = nn.Parameter(torch.zeros((1, c, 1, 1)))


class ConditionalNAFNet(_ImageNAFNet):
    def __init__(self, img_channel=3, width=16, middle_blk_num=1, enc_blk_nums=[], dec_blk_nums=[], upscale=1):
        nn.Module.__init__(self)
        self.upscale = upscale
        self.in_nc = self.out_nc = img_channel
        self.width = width
        self.enc_blk_nums, self.dec_blk_nums, self.middle_blk_num = list(enc_blk_nums), list(dec_blk_nums), middle_blk_num
        time_dim = width * 4
        self.time_mlp = nn.Sequential(nn.Linear(width, time_dim * 2), nn.Identity(), nn.Linear(time_dim, time_dim))
        self.cam_mlp = nn.Sequential(nn.Linear(width * 3, time_dim * 2), nn.Identity(), nn.Linear(time_dim, time_dim))
        self.intro = nn.Conv2d(img_channel * 2, width, 3, padding=1)
        self.ending = nn.Conv2d(width, img_channel, 3, padding=1)
        self.encoders, self.decoders = nn.ModuleList(), nn.ModuleList()
        self.ups, self.downs = nn.ModuleList(), nn.ModuleList()
        chan = width
        for num in self.enc_blk_nums:
            self.encoders.append(nn.Sequential(*[_LensNAFBlock(chan, time_dim) for _ in range(num)]))
            self.downs.append(nn.Conv2d(chan, 2 * chan, 2, 2))
            chan *= 2
        self.middle_blks = nn.Sequential(*[_LensNAFBlock(chan, time_dim) for _ in range(middle_blk_num)])
        for num in self.dec_blk_nums:
            self.ups.append(nn.Sequential(nn.Conv2d(chan, chan * 2, 1, bias=False), nn.Identity()))
            chan //= 2
            self.decoders.append(nn.Sequential(*[_LensNAFBlock(chan, time_dim) for _ in range(num)]))
        self.padder_size = 2 ** len(self.encoders)
        self._engine = None
        self._engine_key = None
        self.engine_flags = 0

    def _create_handle(self, L, device_index, flags):
        return super()._create_handle(L, device_index, flags | _lib.FLAG_NAF_LENS)

    def set_lens_info(self, lens_info, batch, device=None):
        """lens_info = [src_lens, tgt_lens, disparity], each a float or a tensor / sequence with 1 or `batch` entries."""
        if lens_info is None or len(lens_info) != 3:
            raise _lib.IrsdeError("lens_info must be [src_lens, tgt_lens, disparity]")
        cols = []
        for v in lens_info:
            t = torch.as_tensor(v, dtype=torch.float32).reshape(-1).cpu()
            if t.numel() not in (1, batch):
                raise _lib.IrsdeError("each lens_info entry needs 1 or B values")
            cols.append(t.expand(batch) if t.numel() == 1 else t)
        info = torch.stack(cols, dim=1).contiguous()  # [B][3]
        eng = self.engine(device)
        with torch.cuda.device(next(self.parameters()).device if device is None else device):
            _lib.check(_lib.lib().irsde_set_lens_info(eng.h, ctypes.cast(info.data_ptr(), ctypes.POINTER(ctypes.c_float)), batch))

    def forward(self, inp, cond, time, **kwargs):
        if "lens_info" not in kwargs:
            raise _lib.IrsdeError("latent-bokeh ConditionalNAFNet.forward needs lens_info=[src_lens, tgt_lens, disparity]")
        self.set_lens_info(kwargs["lens_info"], inp.shape[0], inp.device)
        return super().forward(inp, cond, time)
