"""Batch sharding of the sampler over the GPUs of one node.

Every image's reverse trajectory is independent (no cross-batch op in the score network), so the
batch is split over ranks with NO collective inside the T-step loop; one all_gather of the final
[B,3,H,W] tensors (RCCL over xGMI when the backend is "nccl", gloo in the CPU tests) reassembles the batch.
Noise streams are keyed by the GLOBAL image index (`sde.image_offset`), so the gathered result does
not depend on the number of ranks — to fp32 summation-order level, not bit for bit: the engine plans per
local batch size (GEMM tilings, split-K, and the chunking of the LinearAttention softmax / context partials
all depend on it), so an image sampled in a shard of 2 and in a batch of 16 differs by ~1e-6 relative per
network evaluation (tests: batch-plan invariance at 5e-5, `tests/test_gpu_fullres.py`).
"""
import torch
import torch.distributed as dist


def shard_bounds(n_items, world_size, rank):
    """Contiguous shard [lo, hi) of `n_items` for `rank`; the first (n_items % world_size) ranks get one
    extra item."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_batch(local, n_items, group=None):
    """all_gather the per-rank shards (shape [n_r, ...]) back into [n_items, ...] on every rank."""
    if not dist.is_available() or not dist.is_initialized():
        return local
    world = dist.get_world_size(group)
    sizes = [shard_bounds(n_items, world, r) for r in range(world)]
    nmax = max(hi - lo for lo, hi in sizes)
    # gloo (CPU tests, or ranks sharing one GPU) gathers host tensors; nccl (= RCCL) gathers in place over xGMI
    stage = local.device if dist.get_backend(group) != "gloo" else torch.device("cpu")
    pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=stage)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0).to(local.device)


def sample_shard(sde, mode, x_local, mu_local, lo, n_items, group=None, **kwargs):
    """Run `sde.reverse_<mode>` on THIS rank's shard (images [lo, lo + len(x_local)) of a global batch of `n_items`)
    and all_gather the restored global batch.  Only the shard has to be resident on the rank.  The noise streams are
    keyed by the global image index (`sde.image_offset` is set to its current value + lo for the call), so the result
    equals one single-GPU call on the whole batch up to fp32 summation order (see the module docstring).  This is the one N>1 sampling path: `sample_sharded` and
    `bench.py --gpus N` go through it (`tools/eval_folder.py --gpus N` shards its FILE LIST over the ranks instead and only
    all_reduces the metric sums)."""
    fn = {"sde": sde.reverse_sde, "ode": sde.reverse_ode, "posterior": sde.reverse_posterior}[mode]
    old_off, old_mu = sde.image_offset, getattr(sde, "mu", None)
    local, err = None, None
    try:
        sde.image_offset = old_off + lo
        sde.set_mu(mu_local)
        local = fn(x_local, **kwargs) if x_local.shape[0] > 0 else x_local.clone()
    except Exception as e:  # noqa: BLE001 - re-raised below, on EVERY rank, after the ranks have agreed
        err = e
    finally:
        sde.image_offset = old_off
        sde.set_mu(old_mu)
    # A rank whose sampler raised (engine error, or the fp16-operand modes' range check `sde._check_fp16_range`) must not
    # leave the others waiting in the all_gather until the RCCL timeout: agree on success first, then raise everywhere.
    bad = all_failed(err is not None, x_local.device, group)
    if err is not None:
        raise err
    if bad:
        raise RuntimeError("sample_shard: the sampler failed on rank(s) %s of this group (their own exception "
                           "is raised there); no rank entered the gather" % bad)
    return gather_batch(local, n_items, group)


def all_failed(failed, device, group=None):
    """Collective: every rank passes its own `failed` flag and gets the sorted list of ranks that failed (empty = all
    fine).  One all_gather of a single int per rank; a no-op list ([] or [0]) without an initialised process group."""
    if not dist.is_available() or not dist.is_initialized():
        return [0] if failed else []
    world = dist.get_world_size(group)
    stage = device if dist.get_backend(group) != "gloo" else torch.device("cpu")
    mine = torch.tensor([1 if failed else 0], dtype=torch.int32, device=stage)
    flags = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(flags, mine, group=group)
    return [r for r, f in enumerate(flags) if int(f.item()) != 0]


def sample_sharded(sde, mode, x_T, mu, group=None):
    """Run `sde.reverse_<mode>` on this rank's shard of (x_T, mu) and gather the full batch.

    x_T / mu are the FULL batch (identical on every rank, e.g. synthetic or broadcast inputs); each rank
    samples images [lo, hi).  Equivalent to one single-GPU call on the full batch."""
    n = x_T.shape[0]
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    lo, hi = shard_bounds(n, world, rank)
    old_noise = sde.injected_noise
    try:
        if old_noise is not None:
            sde.injected_noise = old_noise[:, lo:hi].contiguous()
        sde.set_mu(mu)  # what sample_shard restores afterwards
        return sample_shard(sde, mode, x_T[lo:hi], mu[lo:hi], lo, n, group)
    finally:
        sde.injected_noise = old_noise
