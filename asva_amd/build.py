"""Builds libavsd_hip.so (gfx950 only) in-tree with hipcc.

`python -m asva_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles without a GPU.
Objects are cached under asva_amd/csrc/_obj keyed by source mtime, so rebuilds are incremental.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libavsd_hip.so")
# two builds of the same sources: bfloat16 storage (default) and IEEE-half storage (-DAVSD_F16=1), asva_amd/precision.py
VARIANTS = {"bf16": ("", LIB, []), "fp16": ("_f16", os.path.join(HERE, "libavsd_hip_f16.so"), ["-DAVSD_F16=1"])}
SOURCES = ["lib.hip", "attention.hip", "attention_x2.hip", "gemm_f32.hip", "gemm4.hip", "nstream.hip", "conv3r.hip", "norm.hip", "groupnorm_fused.hip", "elementwise.hip", "audio.hip", "xattn.hip", "attention_fp8.hip", "plan.hip"]
# gemm.hip instantiates ~290 kernels: compiled as six translation units (one per A-loader mode, the entry point, and two for
# the split-precision tiles)
GEMM_UNITS = 7
HEADERS = [os.path.join(CSRC, "avsd_common.h"), os.path.join(CSRC, "gemm_common.h"), os.path.join(CSRC, "gemm4_loops.inc"),
           os.path.join(HERE, "..", "include", "avsd.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


PLAN_HOST = os.path.join(HERE, "plan_host")


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(verbose: bool = False, force: bool = False, variants=("bf16", "fp16")) -> str:
    hipcc = _hipcc()
    jobs = []
    links = []
    for var in variants:
        suffix, lib, defs = VARIANTS[var]
        objdir = OBJ + suffix
        os.makedirs(objdir, exist_ok=True)
        objs = []
        stale = False
        for src in SOURCES:
            s = os.path.join(CSRC, src)
            o = os.path.join(objdir, src.replace(".hip", ".o"))
            objs.append(o)
            if force or _stale(o, [s] + HEADERS):
                jobs.append([hipcc, *FLAGS, *defs, "-c", s, "-o", o])
                stale = True
        gsrc = os.path.join(CSRC, "gemm.hip")
        for u in range(GEMM_UNITS):
            o = os.path.join(objdir, f"gemm_tu{u}.o")
            objs.append(o)
            if force or _stale(o, [gsrc] + HEADERS):
                jobs.insert(0, [hipcc, *FLAGS, *defs, f"-DAVSD_GEMM_TU={u}", "-c", gsrc, "-o", o])     # longest jobs first
                stale = True
        links.append((lib, objs, stale))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(7, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    for lib, objs, stale in links:
        if force or stale or _stale(lib, objs):
            run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
    # the Python-free host of the launch plans (include/avsd.h "launch plans"): dlopens either library
    host_src = os.path.join(HERE, "..", "tools", "plan_host.cpp")
    if os.path.exists(host_src) and (force or _stale(PLAN_HOST, [host_src] + HEADERS)):
        run([hipcc, "-O2", "-std=c++17", "-I", os.path.join(HERE, "..", "include"), host_src, "-ldl", "-o", PLAN_HOST])
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
