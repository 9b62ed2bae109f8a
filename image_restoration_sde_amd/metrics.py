"""Evaluation tail on the device — the metric block of the reference's test scripts
(/root/reference/codes/config/deraining/test.py:110-178) without cv2 and without leaving the GPU:
`util.tensor2img`, `util.calculate_psnr`, `util.calculate_ssim` (codes/utils/img_utils.py:136-234) and the Y-channel
variants through `bgr2ycbcr` (codes/data/util.py:177-198).  All arithmetic runs in libirsde_hip.so
(csrc/eval_metrics.hip); there is no CPU fallback.
"""
import ctypes

import numpy as np
import torch

from . import _lib


def _f32_cuda(t):
    if not isinstance(t, torch.Tensor) or t.device.type != "cuda":
        raise _lib.IrsdeError("metrics run on CUDA(HIP) tensors only (no CPU fallback)")
    t = t.detach().to(torch.float32)
    if t.dim() == 3:
        t = t.unsqueeze(0)
    if t.dim() != 4:
        raise _lib.IrsdeError("expected (B,C,H,W) or (C,H,W)")
    return t.contiguous()


def tensor2img(tensor, out_type=np.uint8, min_max=(0, 1)):
    """util.tensor2img for a (C,H,W) / (1,C,H,W) device tensor: numpy HWC uint8, BGR channel order."""
    if out_type != np.uint8 or tuple(min_max) != (0, 1):
        raise _lib.IrsdeError("only the uint8 / (0,1) form used by the test scripts is implemented")
    t = _f32_cuda(tensor)
    B, C, H, W = t.shape
    if B != 1:
        raise _lib.IrsdeError("tensor2img takes one image (use tensor2img_batch)")
    return tensor2img_batch(t)[0]


def tensor2img_batch(tensor):
    """(B,C,H,W) device tensor -> numpy (B,H,W,C) uint8 BGR (C == 1: (B,H,W))."""
    t = _f32_cuda(tensor)
    B, C, H, W = t.shape
    out = torch.empty((B, H, W, C), dtype=torch.uint8, device=t.device)
    with torch.cuda.device(t.device):
        _lib.check(_lib.lib().irsde_tensor2img(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(out.data_ptr()), B, C, H, W,
                                               _lib.stream_ptr()))
    a = out.cpu().numpy()
    return a[..., 0] if C == 1 else a


def evaluate_batch(output, gt, crop_border=0):
    """Per-image metrics of a batch: dict of float64 arrays psnr, ssim, psnr_y, ssim_y (deraining/test.py:137-178;
    crop_border as `opt["crop_border"]`)."""
    o, g = _f32_cuda(output), _f32_cuda(gt)
    if o.shape != g.shape or o.device != g.device:
        raise _lib.IrsdeError("output / GT shape or device mismatch")
    B, C, H, W = o.shape
    m = (ctypes.c_double * (4 * B))()
    with torch.cuda.device(o.device):
        _lib.check(_lib.lib().irsde_eval_metrics(ctypes.c_void_p(o.data_ptr()), ctypes.c_void_p(g.data_ptr()), B, C, H, W,
                                                 int(crop_border), m, _lib.stream_ptr()))
    a = np.array(m, dtype=np.float64).reshape(B, 4)
    return {"psnr": a[:, 0], "ssim": a[:, 1], "psnr_y": a[:, 2], "ssim_y": a[:, 3]}


def reduce_metrics(metrics):
    """Dataset averages across ranks: all_reduce of (sum, count) per metric (the optional scalar
    reduction); single-process when torch.distributed is not initialised."""
    import torch.distributed as dist
    keys = sorted(metrics)
    local = torch.tensor([[float(np.sum(metrics[k])), float(len(metrics[k]))] for k in keys], dtype=torch.float64)
    if dist.is_available() and dist.is_initialized():
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        buf = local.to(dev)
        dist.all_reduce(buf)
        local = buf.cpu()
    return {k: float(local[i, 0] / max(local[i, 1], 1.0)) for i, k in enumerate(keys)}
