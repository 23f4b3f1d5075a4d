import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    # The CPU oracle runs on torch's CPU convolutions, which do not scale to every core of a large host: on the 128-core GPU box one
    # evaluation takes 18x longer at 128 threads than at 16 (bench.py's thread scan) -- the two full-size sampler tests alone took
    # 290 s of a 616 s suite before this cap.
    try:
        import torch
        if torch.get_num_threads() > 16:
            torch.set_num_threads(16)
    except Exception:
        pass
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "compat_branch: a gpu test that exercises the interface-compatibility branch on purpose")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "hotpath_golden.npz"))


@pytest.fixture(autouse=True)
def _device_loop_is_what_runs(request, monkeypatch):
    """Every ``-m gpu`` test runs with the interface-compatibility branches of the plugins turned into errors, and every
    ``NativeHandle.sampler_run`` is checked against the library's own counters (``adf_get_counters``): the call must have
    been served by ``adf_sampler_run`` -- one more run, its evaluations, and a hipGraph replay exactly when the descriptor
    asks for one.  A refactor that sends the package's own (fn, net) pair down the tensor-op restatement of a sampler
    (which would still call the HIP net per evaluation and reproduce the reference) fails here instead of staying green."""
    if request.node.get_closest_marker("gpu") is None or request.node.get_closest_marker("compat_branch") is not None:
        yield
        return
    import audiodiffuser_amd.samplers as S
    from audiodiffuser_amd.net import NativeHandle

    monkeypatch.setenv("ADF_REQUIRE_NATIVE", "1")
    monkeypatch.setattr(S, "REQUIRE_NATIVE", True)
    real = NativeHandle.sampler_run
    seen = {"runs": 0}

    def checked(self, desc, sigmas_host, noise, injected):
        before = self.counters()
        out = real(self, desc, sigmas_host, noise, injected)
        after = self.counters()
        nfe = self.lib.adf_sampler_nfe(desc, (__import__("ctypes").c_float * sigmas_host.numel())(*sigmas_host.detach().cpu().float().tolist()),
                                       sigmas_host.numel())
        assert after["sampler_runs"] == before["sampler_runs"] + 1, "adf_sampler_run did not complete a device loop"
        assert nfe > 0 and after["sampler_evals"] == before["sampler_evals"] + nfe, (before, after, nfe)
        replayed = after["graph_replays"] - before["graph_replays"]
        assert replayed == (1 if desc.use_graph else 0), f"use_graph={desc.use_graph} but {replayed} graph replays"
        seen["runs"] += 1
        return out

    monkeypatch.setattr(NativeHandle, "sampler_run", checked)
    request.node._adf_seen = seen
    yield
