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


def test_launch_ranks_path(tmp_path):
    """The launcher bench.py --gpus N uses (audiodiffuser_amd.distributed.launch_ranks -> torch.distributed.run, rendezvous on
    127.0.0.1) with 2 CPU ranks over gloo: both ranks come up, the gathered result equals the unsharded run."""
    import json
    import subprocess
    from audiodiffuser_amd.distributed import launch_ranks
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "diag", "dist_launcher_probe.py")
    log = tmp_path / "out.txt"
    with open(log, "w") as f:
        rc = launch_ranks(probe, 2, ["6", "128"], stdout=f, timeout=300)
    assert rc == 0
    line = [l for l in open(log).read().splitlines() if l.startswith("{")][-1]
    rep = json.loads(line)
    assert rep["n_ranks"] == 2 and rep["env_world"] == 2 and rep["master"] == "127.0.0.1" and rep["shape"] == [6, 1, 128]
    want = _run_local(generate_noise(0, 6, 128))
    assert abs(rep["sum"] - float(want.double().sum())) < 1e-9 * max(1.0, abs(float(want.double().sum())))


def test_bench_refuses_more_gpus_than_devices():
    """bench.py --gpus N must not silently run on fewer devices: with no (or too few) GPUs visible it exits non-zero."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "--gpus 64" in r.stderr


def _json_lines(text):
    return [l for l in text.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("how", ["torchrun", "self_launch"])
def test_bench_rank_path_prints_one_line_from_rank_0(how):
    """bench.py's rank path (rendezvous on 127.0.0.1, index-keyed noise slices, fences, MAX over ranks, ONE all-gather, rank 0
    reporting) with the device layer mocked (`--mock-device`: CPU tensors, gloo, a stand-in step): exactly one JSON line,
    n_gpus == rccl_ranks == WORLD_SIZE, the gathered batch is the global one -- started the way the driver starts it
    (torch.distributed.run) and the way `bench.py --gpus N` starts itself."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(root, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    args = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--mock-device", "--batch", "3", "--length", "128", "--config", "c3", "--sampler", "dpm"]
    if how == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), bench] + args
    else:
        cmd = [sys.executable, bench] + args
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == j["rccl_ranks"] == 2 and j["steps"] == 2 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["mock_device"] is True and j["value"] is None                       # a rehearsal never reports a throughput
    assert j["config"]["global_batch"] == 6 and j["gathered_shape"] == [6, 1, 128] and j["nfe_per_waveform"] == 49
    want = torch.tanh(generate_noise(0, 6, 128) * 0.5)                            # sharding does not change any sample
    assert abs(j["gathered_checksum"] - float(want.double().sum())) < 1e-9 * max(1.0, abs(float(want.double().sum())))
    # a mismatch between --gpus and the launcher's world size is refused by every rank (exit 2), nothing is printed
    if how == "torchrun":
        bad = [a if a != "2" else "4" for a in cmd]
        bad[bad.index("--nproc-per-node") + 1] = "2"
        r = subprocess.run(bad, capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode != 0 and not _json_lines(r.stdout)
