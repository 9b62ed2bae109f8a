"""N>1 path on CPU: world_size-2 gloo run of the batch sharding + final gather.

The GPU sampler itself cannot run here, so the `sde` object is a stand-in whose per-image result
depends on (x, mu, global image index, injected noise) exactly like the real sampler's contract;
what is tested is `sample_sharded` / `gather_batch` / `shard_bounds`: every rank ends with the same
full batch a single process computes, for even and ragged splits."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import image_restoration_sde_amd as P


class FakeSDE:
    def __init__(self):
        self.image_offset = 0
        self.injected_noise = None
        self.mu = None
        self.calls = 0

    def set_mu(self, mu):
        self.mu = mu

    def _f(self, x):
        self.calls += 1
        idx = torch.arange(x.shape[0], dtype=x.dtype).view(-1, 1, 1, 1) + self.image_offset
        out = 2 * x + self.mu + 0.01 * idx
        if self.injected_noise is not None:
            out = out + self.injected_noise[1:].sum(0)
        return out

    reverse_sde = reverse_ode = reverse_posterior = _f


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(3)
        x = torch.randn(n, 3, 4, 5, generator=g)
        mu = torch.rand(n, 3, 4, 5, generator=g)
        z = torch.randn(4, n, 3, 4, 5, generator=g)
        sde = FakeSDE()
        sde.injected_noise = z
        out = P.sample_sharded(sde, "posterior", x, mu)
        assert sde.image_offset == 0 and sde.injected_noise is z  # restored
        q.put((rank, out.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [4, 5])
def test_sample_sharded_world2_gloo(n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, 3, 4, 5, generator=g)
    mu = torch.rand(n, 3, 4, 5, generator=g)
    z = torch.randn(4, n, 3, 4, 5, generator=g)
    ref = FakeSDE()
    ref.injected_noise = z
    ref.set_mu(mu)
    want = ref.reverse_posterior(x).numpy()
    assert np.array_equal(res[0], want) and np.array_equal(res[1], want)


@pytest.mark.parametrize("n", [16, 13, 5])
def test_sample_sharded_world8_gloo(n):
    """The real world size of the node (8 ranks): BASELINE configs[1]'s 16 images (2 per rank), a ragged 13 (five ranks with 2, three with 1)
    and 5 images (three ranks with an EMPTY shard: they skip the sampler and still take part in the gather)."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, 3, 4, 5, generator=g)
    mu = torch.rand(n, 3, 4, 5, generator=g)
    z = torch.randn(4, n, 3, 4, 5, generator=g)
    ref = FakeSDE()
    ref.injected_noise = z
    ref.set_mu(mu)
    want = ref.reverse_posterior(x).numpy()
    for r in range(world):
        assert np.array_equal(res[r], want), r
    bounds = [P.dist.shard_bounds(n, world, r) for r in range(world)]
    assert bounds[0][0] == 0 and bounds[-1][1] == n and all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1))
    assert max(hi - lo for lo, hi in bounds) - min(hi - lo for lo, hi in bounds) <= 1


class FailingSDE(FakeSDE):
    """Stand-in for a rank whose sampler raises (engine error / the fp16-operand modes' range check)."""
    def _f(self, x):
        raise FloatingPointError("non-finite sampler output on this rank")

    reverse_sde = reverse_ode = reverse_posterior = _f


def _failing_worker(rank, world, port, bad_rank, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(3)
        x = torch.randn(4, 3, 4, 5, generator=g)
        mu = torch.rand(4, 3, 4, 5, generator=g)
        sde = FailingSDE() if rank == bad_rank else FakeSDE()
        try:
            P.sample_sharded(sde, "sde", x, mu)
            q.put((rank, "returned"))
        except FloatingPointError:
            q.put((rank, "own"))
        except RuntimeError as e:
            q.put((rank, "peer:" + str(e)))
        assert sde.image_offset == 0
        # the group is still usable afterwards: nobody is stuck inside a half-entered gather
        t = torch.ones(1)
        dist.all_reduce(t)
        assert int(t.item()) == world
    finally:
        dist.destroy_process_group()


def test_sample_shard_failure_on_one_rank_raises_on_all_ranks():
    """a rank whose sampler raises must not leave the others waiting in the all_gather.  Every rank agrees on
    success BEFORE the gather: the failing rank re-raises its own exception, the others raise a RuntimeError naming it."""
    world, bad = 2, 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port, bad, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[bad] == "own"
    assert res[0].startswith("peer:") and "[1]" in res[0]


def test_single_process_passthrough():
    sde = FakeSDE()
    x = torch.ones(3, 3, 2, 2)
    out = P.sample_sharded(sde, "sde", x, x)
    assert out.shape == x.shape and sde.calls == 1


def _metrics_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank 0 evaluated 3 images, rank 1 two: the dataset average is the count-weighted mean, not the mean of means
        local = {"psnr": np.array([30.0, 31.0, 32.0]) if rank == 0 else np.array([20.0, 21.0]),
                 "ssim": np.array([0.9, 0.8, 0.7]) if rank == 0 else np.array([0.5, 0.6])}
        q.put((rank, P.metrics.reduce_metrics(local)))
    finally:
        dist.destroy_process_group()


def test_reduce_metrics_world2_gloo():
    """The optional scalar reduction of the evaluation tail: sum/count all_reduce across ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_metrics_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        assert abs(res[r]["psnr"] - (30 + 31 + 32 + 20 + 21) / 5) < 1e-12
        assert abs(res[r]["ssim"] - (0.9 + 0.8 + 0.7 + 0.5 + 0.6) / 5) < 1e-12
    assert P.metrics.reduce_metrics({"psnr": np.array([1.0, 3.0])})["psnr"] == 2.0  # single process
