"""`ConditionalNAFNet` — the Refusion score network's interface on top of the HIP engine.

Mirrors `ConditionalNAFNet(img_channel=3, width=16, middle_blk_num=1, enc_blk_nums=[], dec_blk_nums=[], upscale=1)` and
`forward(inp, cond, time)` of /root/reference/codes/config/deraining/models/modules/DenoisingNAFNet_arch.py:85-187, so
`which_model_G: ConditionalNAFNet` (deraining/options/test/refusion.yml:29-35) and reference checkpoints load unchanged.
Sub-modules only own parameters under the reference's state_dict names; all arithmetic runs in libirsde_hip.so
(1x1 convolutions on the fp32 MFMA implicit-GEMM kernel with the SimpleGate / SCA / PixelShuffle / beta-gamma residual
fusions, depthwise 3x3 + gate + pooled-sum kernel, LayerNorm + FiLM kernel).  No PyTorch fallback.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib
from .unet import ConditionalUNet, _Gain


class _NAFBlock(nn.Module):  # DenoisingNAFNet_arch.py:15-49 (parameter container)
    def __init__(self, c, time_emb_dim):
        super().__init__()
        self.mlp = nn.Sequential(nn.Identity(), nn.Linear(time_emb_dim // 2, c * 4))
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


class ConditionalNAFNet(ConditionalUNet):
    """Shares engine management / forward with ConditionalUNet (same C entry points after creation)."""

    def __init__(self, img_channel=3, width=16, middle_blk_num=1, enc_blk_nums=[], dec_blk_nums=[], upscale=1):
        nn.Module.__init__(self)
        self.upscale = upscale
        self.in_nc = self.out_nc = img_channel
        self.width = width
        self.enc_blk_nums, self.dec_blk_nums, self.middle_blk_num = list(enc_blk_nums), list(dec_blk_nums), middle_blk_num
        time_dim = width * 4
        self.time_mlp = nn.Sequential(nn.Identity(), nn.Linear(width, time_dim * 2), nn.Identity(), nn.Linear(time_dim, time_dim))
        self.intro = nn.Conv2d(img_channel * 2, width, 3, padding=1)
        self.ending = nn.Conv2d(width, img_channel, 3, padding=1)
        self.encoders, self.decoders = nn.ModuleList(), nn.ModuleList()
        self.ups, self.downs = nn.ModuleList(), nn.ModuleList()
        chan = width
        for num in self.enc_blk_nums:
            self.encoders.append(nn.Sequential(*[_NAFBlock(chan, time_dim) for _ in range(num)]))
            self.downs.append(nn.Conv2d(chan, 2 * chan, 2, 2))
            chan *= 2
        self.middle_blks = nn.Sequential(*[_NAFBlock(chan, time_dim) for _ in range(middle_blk_num)])
        for num in self.dec_blk_nums:
            self.ups.append(nn.Sequential(nn.Conv2d(chan, chan * 2, 1, bias=False), nn.Identity()))
            chan //= 2
            self.decoders.append(nn.Sequential(*[_NAFBlock(chan, time_dim) for _ in range(num)]))
        self.padder_size = 2 ** len(self.encoders)
        self._engine = None
        self._engine_key = None
        self.engine_flags = 0

    def _create_handle(self, L, device_index, flags):
        cfg = _lib.NafConfig()
        cfg.img_channel, cfg.width, cfg.middle_blk_num = self.in_nc, self.width, self.middle_blk_num
        cfg.n_enc, cfg.n_dec = len(self.enc_blk_nums), len(self.dec_blk_nums)
        for i, v in enumerate(self.enc_blk_nums):
            cfg.enc_blk_nums[i] = v
        for i, v in enumerate(self.dec_blk_nums):
            cfg.dec_blk_nums[i] = v
        cfg.device, cfg.flags = device_index, flags
        h = ctypes.c_void_p()
        _lib.check(L.irsde_create_nafnet(ctypes.byref(cfg), ctypes.byref(h)))
        return h
