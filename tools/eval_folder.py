"""eval_folder.py — the dataset evaluation loop of the reference's test scripts on the MI355X sampler.

Reproduces /root/reference/codes/config/deraining/test.py:93-217 on the product API: walk an LQ (and optional GT)
image folder -> `sde.noise_state(LQ)` -> `model.feed_data / model.test(sde, mode) / get_current_visuals` semantics
(here: batched `sde.reverse_*` calls) -> `tensor2img` -> save PNG -> PSNR / SSIM on RGB and on the Y channel ->
dataset averages.  This is the one command for the Rain100H gate of BASELINE.json (PSNR within 0.05 dB of the
reference's 31.65 dB, /root/reference/README.md:42-46) once `rain100h_sde.pth` and the dataset are supplied:

    python tools/eval_folder.py --lq Rain100H/LQ --gt Rain100H/GT --weights rain100h_sde.pth \
        --max-sigma 10 --T 100 --mode posterior --scale 4 --out results/Rain100H [--gpus 8]

(`--scale 4` is the default: the reference crops `crop_border or scale` = 4 pixels before PSNR / SSIM, test.py:134.)

Differences from the reference loop, all stated: images of equal size are sampled as one batch (`--batch`); with
`--gpus N` the image list is sharded over N ranks (one process per GPU, no collective except the final metric
all_reduce); image I/O is PIL instead of cv2 (PNG decoding is lossless, so the tensors are identical; JPEG decoders
may differ in the last bit); LPIPS is not computed (the `lpips` package and its AlexNet weights are external
downloads); the noise of `noise_state` is drawn per image from a generator seeded by (--seed, image index) so a run is
reproducible and independent of batching / sharding.
"""
import argparse
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

IMG_EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".tif")  # codes/data/util.py:12


def list_images(folder):
    """Sorted image paths of a folder tree (codes/data/util.py:17-28 `_get_paths_from_images`)."""
    out = []
    for dirpath, _, fnames in sorted(os.walk(folder)):
        for f in sorted(fnames):
            if f.lower().endswith(IMG_EXTENSIONS):
                out.append(os.path.join(dirpath, f))
    if not out:
        raise FileNotFoundError("%s has no valid image file" % folder)
    return out


def pair_paths(lq_dir, gt_dir):
    """[(lq_path, gt_path | None)] in sorted order; like LQGTDataset the i-th LQ file goes with the i-th GT file
    (codes/data/LQGT_dataset.py:48-54), and the counts must agree."""
    lq = list_images(lq_dir)
    if gt_dir is None:
        return [(p, None) for p in lq]
    gt = list_images(gt_dir)
    if len(gt) != len(lq):
        raise ValueError("GT and LQ datasets have different number of images - %d, %d" % (len(gt), len(lq)))
    return list(zip(lq, gt))


def read_img(path):
    """codes/data/util.py:65-81 `read_img` + the BGR->RGB / HWC->CHW of LQGT_dataset.py:177-186: float32 CHW RGB
    in [0, 1]."""
    import numpy as np
    from PIL import Image
    im = Image.open(path)
    a = np.asarray(im.convert("RGB") if im.mode != "L" else im, dtype=np.float32) / np.float32(255.0)
    if a.ndim == 2:
        a = a[:, :, None]
    return np.ascontiguousarray(a.transpose(2, 0, 1))


def save_img(img_bgr, path):
    """util.save_img (cv2.imwrite of a BGR / gray uint8 image) through PIL."""
    from PIL import Image
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    Image.fromarray(img_bgr[..., ::-1] if img_bgr.ndim == 3 else img_bgr).save(path)


def batches_of_equal_size(items, sizes, max_batch):
    """Consecutive-in-order grouping: indices into `items` batched while the (H, W) stays the same."""
    out, cur = [], []
    for i in range(len(items)):
        if cur and (sizes[i] != sizes[cur[0]] or len(cur) >= max_batch):
            out.append(cur)
            cur = []
        cur.append(i)
    if cur:
        out.append(cur)
    return out


def build_model(a, P, torch):
    if a.model == "nafnet":
        m = P.ConditionalNAFNet(img_channel=3, width=64, enc_blk_nums=[1, 1, 1, 28], middle_blk_num=1, dec_blk_nums=[1, 1, 1, 1])
    else:
        m = P.ConditionalUNet(3, 3, a.nf, depth=a.depth)
    sd = torch.load(a.weights, map_location="cpu")
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}  # base_model.py:92-105
    m.load_state_dict(sd, strict=True)
    return m


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--lq", required=True)
    ap.add_argument("--gt", default=None)
    ap.add_argument("--weights", required=True, help="reference checkpoint (state_dict .pth, optional 'module.' prefixes)")
    ap.add_argument("--out", default=None, help="results folder: <name>.png, <name>_LQ.png, <name>_HQ.png as test.py:118-128")
    ap.add_argument("--model", default="unet", choices=["unet", "nafnet"])
    ap.add_argument("--nf", type=int, default=64)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--max-sigma", type=float, default=10)
    ap.add_argument("--T", type=int, default=100)
    ap.add_argument("--schedule", default="cosine")
    ap.add_argument("--eps", type=float, default=0.005)
    ap.add_argument("--mode", default="posterior", choices=["sde", "ode", "posterior"])
    ap.add_argument("--crop-border", type=int, default=None, help="opt['crop_border']; unset / 0 falls back to --scale exactly as test.py:134 does")
    ap.add_argument("--scale", type=int, default=4, help="opt['degradation']['scale'] (deraining options/test/ir-sde.yml:20 sets 4 and no crop_border, "
                    "so the published Rain100H PSNR / SSIM are computed on images cropped by 4 px)")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16", "bf16_act", "fp16"])
    a = ap.parse_args(argv)
    crop_border = a.crop_border if a.crop_border else a.scale   # test.py:134: `opt["crop_border"] if opt["crop_border"] else scale`

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
        return subprocess.call(cmd)

    import numpy as np
    import torch
    import torch.distributed as dist
    import image_restoration_sde_amd as P

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit("eval_folder.py: --gpus %d does not match WORLD_SIZE=%d" % (a.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    pairs = pair_paths(a.lq, a.gt)
    lo, hi = P.shard_bounds(len(pairs), world, rank)   # this rank's images; noise keyed by the global index
    model = build_model(a, P, torch).to(dev).eval()
    if a.dtype != "fp32":
        model.set_compute_dtype(a.dtype)
    sde = P.IRSDE(max_sigma=a.max_sigma, T=a.T, schedule=a.schedule, eps=a.eps, device=dev)
    sde.set_model(model)
    sde.seed = a.seed

    mine = list(range(lo, hi))
    lq_np = [read_img(pairs[i][0]) for i in mine]
    sizes = [x.shape for x in lq_np]
    res = {"psnr": [], "ssim": [], "psnr_y": [], "ssim_y": []}
    times = []
    log = (lambda *s: print(*s, flush=True))
    for grp in batches_of_equal_size(mine, sizes, a.batch):
        LQ = torch.from_numpy(np.stack([lq_np[j] for j in grp]))
        noise = torch.stack([torch.randn(LQ.shape[1:], generator=torch.Generator().manual_seed(a.seed * 1000003 + mine[j]))
                             for j in grp])
        noisy_state = LQ + noise * sde.max_sigma           # IRSDE.noise_state (sde_utils.py:360-361), on the CPU as test.py:104
        sde.image_offset = mine[grp[0]]
        sde.set_mu(LQ.to(dev))
        torch.cuda.synchronize()
        tic = time.time()
        fn = {"sde": sde.reverse_sde, "ode": sde.reverse_ode, "posterior": sde.reverse_posterior}[a.mode]
        out = fn(noisy_state.to(dev))
        torch.cuda.synchronize()
        times.append((time.time() - tic) / len(grp))
        names = [os.path.splitext(os.path.basename(pairs[mine[j]][1] or pairs[mine[j]][0]))[0] for j in grp]
        GT = None
        if a.gt is not None:
            GT = torch.from_numpy(np.stack([read_img(pairs[mine[j]][1]) for j in grp])).to(dev)
            m = P.metrics.evaluate_batch(out, GT, crop_border=crop_border)
            for k in res:
                res[k].extend(m[k].tolist())
            for n, name in enumerate(names):
                log("img%3d:%-15s - PSNR: %.6f dB; SSIM: %.6f; PSNR_Y: %.6f dB; SSIM_Y: %.6f." %
                    (mine[grp[n]], name, m["psnr"][n], m["ssim"][n], m["psnr_y"][n], m["ssim_y"][n]))
        if a.out:
            imgs = P.metrics.tensor2img_batch(out)
            lqi = P.metrics.tensor2img_batch(LQ.to(dev))
            gti = P.metrics.tensor2img_batch(GT) if GT is not None else None
            for n, name in enumerate(names):
                save_img(imgs[n], os.path.join(a.out, name + ".png"))
                save_img(lqi[n], os.path.join(a.out, name + "_LQ.png"))
                if gti is not None:
                    save_img(gti[n], os.path.join(a.out, name + "_HQ.png"))
    summary = None
    if a.gt is not None:
        local = {k: np.asarray(v, dtype=np.float64) for k, v in res.items()}
        if world > 1 or len(local["psnr"]):
            summary = P.metrics.reduce_metrics(local)
    if rank == 0 and summary is not None:
        log("----Average PSNR/SSIM results for %s----\n\tPSNR: %.6f dB; SSIM: %.6f\n" % (os.path.basename(os.path.normpath(a.lq)), summary["psnr"], summary["ssim"]))
        log("----Y channel, average PSNR/SSIM----\n\tPSNR_Y: %.6f dB; SSIM_Y: %.6f\n" % (summary["psnr_y"], summary["ssim_y"]))
    if rank == 0 and times:
        log("average test time per image: %.4f s (rank 0, %d images over %d rank(s))" % (float(np.mean(times)), len(pairs), world))
    if world > 1:
        dist.destroy_process_group()
    return summary


if __name__ == "__main__":
    r = main()
    sys.exit(r if isinstance(r, int) else 0)
