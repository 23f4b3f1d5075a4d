"""A rank of the launcher test (tests/test_dist_gloo.py::test_launch_ranks_path): started by
audiodiffuser_amd.distributed.launch_ranks exactly as bench.py --gpus N starts its ranks, but on the CPU with gloo and a mock
denoiser.  Rank 0 prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import audiodiffuser_amd as A  # noqa: E402
from audiodiffuser_amd.distributed import sample_sharded  # noqa: E402


def run_local(noise):
    mock = lambda x, net=None, sigma=None, **kw: 0.25 * x + 0.1
    sig = A.KarrasSchedule(0.002, 80.0, 7.0, 10)()
    return A.EDMSampler(s_churn=0.0, s_noise=1.0, num_steps=10)(noise, fn=mock, net=None, sigmas=sig)


gb, length = int(sys.argv[1]), int(sys.argv[2])
dist.init_process_group("gloo")
out = sample_sharded(run_local, gb, length, torch.device("cpu"))
if dist.get_rank() == 0:
    print(json.dumps({"n_ranks": dist.get_world_size(), "env_world": int(os.environ["WORLD_SIZE"]), "shape": list(out.shape),
                      "sum": float(out.double().sum()), "master": os.environ.get("MASTER_ADDR")}))
dist.barrier()
dist.destroy_process_group()
