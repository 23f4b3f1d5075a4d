"""Batch-sharded sampling over the GPUs of one node (one process per GPU, RCCL over xGMI).

Every waveform's trajectory is independent (no cross-sample op anywhere in the U-Net; GroupNorm and
LayerNorm are per-sample), so the global batch is cut into contiguous per-rank slices, each rank runs
the whole captured sampler loop on its slice with replicated weights, and the only exchange is ONE
all-gather of the finished waveforms at the end.  The reference has no counterpart:
its inference is unsharded (src/models/diffunet_complex_module.py:235-266).
"""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
from typing import Callable, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .weights import generate_noise


def shard_range(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, stop) slice of the global sample index range owned by ``rank``."""
    base, rem = divmod(global_batch, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def rank_noise(global_batch: int, length: int, rank: int, world: int, channels: int = 1, base_seed: int = 1234) -> torch.Tensor:
    """Initial noise of this rank's slice; sample i always comes from seed base_seed + i."""
    lo, hi = shard_range(global_batch, rank, world)
    return generate_noise(lo, hi - lo, length, channels, base_seed)


def gather_samples(local: torch.Tensor, global_batch: int, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """All-gather the per-rank results into the [global_batch, C, L] tensor on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [shard_range(global_batch, r, world) for r in range(world)]
    if len({hi - lo for lo, hi in sizes}) == 1:
        out = torch.empty((global_batch,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    # ragged split: pad every slice to the largest one (collectives need equal shapes), gather once, trim
    nmax = max(hi - lo for lo, hi in sizes)
    padded = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = torch.empty((world * nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * nmax: r * nmax + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], dim=0)


def sample_sharded(run_local: Callable[[torch.Tensor], torch.Tensor], global_batch: int, length: int, device: torch.device,
                   channels: int = 1, base_seed: int = 1234, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """``run_local(noise_slice) -> samples_slice`` on this rank's slice, then one all-gather."""
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    noise = rank_noise(global_batch, length, rank, world, channels, base_seed).to(device)
    return gather_samples(run_local(noise), global_batch, group)


def launch_ranks(script: str, n: int, argv: Sequence[str], stdout=None, timeout: Optional[float] = None) -> int:
    """Start ``n`` ranks of ``script`` on this node -- one process per GPU -- with ``torch.distributed.run`` and return its
    exit code.  The caller must not have touched a GPU yet (a process that has initialised HIP must not be replaced, and
    the children open the devices themselves); rendezvous is on 127.0.0.1 (the container hostname may not resolve).
    ``bench.py --gpus N`` uses this when it is not already running under a launcher."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: needed by RCCL on this driver
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script] + list(argv)
    # own session: the ranks are grandchildren, so on a timeout the whole process GROUP is ended (they would otherwise keep
    # their GPUs and the rendezvous port); the ranks stay fresh child processes -- nothing that touched a GPU is re-exec'ed
    proc = subprocess.Popen(cmd, env=env, stdout=stdout, start_new_session=True)
    try:
        return proc.wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        return 124
