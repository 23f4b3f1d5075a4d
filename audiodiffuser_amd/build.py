"""Builds ``libadf_hip.so`` (the C-ABI HIP library) in-tree for gfx950 with hipcc.

No GPU is needed to build (hipcc cross-compiles).  The .so sits next to this file so it
travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libadf_hip.so")
SOURCES = ["adf_gemm.hip", "adf_kernels.hip", "adf_wavenet.hip", "adf_conv2d.hip", "adf_api.hip", "adf_net_unet1d.hip", "adf_net_wavenet.hip",
           "adf_net_adm.hip", "adf_sampler.hip", "adf_bench_replay.hip"]
# adf_gemm.hip: the SLP vectoriser would pair the prologue arithmetic that adf_gemm_pp.h places one element per MFMA gap
# into v_pk_* operations (which are slower beside MFMAs and land in one gap instead of two)
EXTRA_FLAGS = {"adf_gemm.hip": ["-fno-slp-vectorize"]}
HEADERS = ["adf_common.h", "adf_gemm.h", "adf_gemm_pp.h", "adf_gemm_rb.h", "adf_gemm_rbx3.h", "adf_gemm_up.h", "adf_kernels.h", "adf_wavenet.h", "adf_conv2d.h", "adf_transformer.h", "adf_resblock_small.h", "adf_resblock_split.h", "adf_api_internal.h", os.path.join("..", "..", "include", "audiodiffuser_amd.h")]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build libadf_hip.so)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    flags = [f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(bdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + flags + EXTRA_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return o

    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(compile_one, jobs))
    objs = [os.path.join(bdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
