"""Build libelvis_amd.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m elvis_amd._build          # incremental: recompiles only stale objects

The shared object lands in elvis_amd/lib/ (git-ignored, but it travels to the GPU box with
the gpurun snapshot).  No torch involvement: the boundary is a plain C ABI.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIBPATH = os.path.join(LIBDIR, "libelvis_amd.so")
SOURCES = ["conv.hip", "conv_f32.hip", "dcn.hip", "swin.hip", "api.hip", "glue.hip", "misc.hip", "norm.hip", "attn.hip", "degrade.hip"]   # slowest first
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result",
         "-ffp-contract=off"]  # bit-exact glue: no implicit FMA contraction (explicit fmaf where wanted)
# per-source extras: the tiled DCNv2 kernel fully unrolls its 64 (group, tap) samples so that every register
# index is static - beyond clang's default size limit for `#pragma unroll`
EXTRA_FLAGS = {"dcn.hip": ["-mllvm", "-pragma-unroll-threshold=262144"]}


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; libelvis_amd cannot be built")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "elvis_amd.h"))
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stderr}")
        if verbose:
            print(f"[elvis_amd] compiled {os.path.basename(s)}", file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJDIR, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIBPATH, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIBPATH] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        if verbose:
            print(f"[elvis_amd] linked {LIBPATH}", file=sys.stderr)
    return LIBPATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
