"""CPU, world_size 2 over gloo: the batch-sharded sampling driver (audiodiffuser_amd/distributed.py).
The per-rank work is the sampler's interface-compatibility branch with a mock denoiser, so the test covers
the N>1 control flow (slice ownership, index-keyed noise, the single all-gather) without a GPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import audiodiffuser_amd as A
from audiodiffuser_amd.distributed import sample_sharded, shard_range
from audiodiffuser_amd.weights import generate_noise


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_local(noise):
    mock = lambda x, net=None, sigma=None, **kw: 0.25 * x + 0.1
    sig = A.KarrasSchedule(0.002, 80.0, 7.0, 10)()
    return A.EDMSampler(s_churn=0.0, s_noise=1.0, num_steps=10)(noise, fn=mock, net=None, sigmas=sig)


def _worker(rank, world, port, gb, length, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = sample_sharded(_run_local, gb, length, torch.device("cpu"))
        q.put((rank, out.clone()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("gb", [6, 5])     # even and ragged split
def test_sharded_sampling_matches_single_process(gb):
    world, length = 2, 128
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, gb, length, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _run_local(generate_noise(0, gb, length))
    for r in range(world):
        assert got[r].shape == want.shape
        assert torch.equal(got[r], want), f"rank {r} gathered result differs from the unsharded run"
    lo, hi = shard_range(gb, 1, world)
    assert hi == gb and lo == (gb + 1) // 2
