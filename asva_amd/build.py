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
SOURCES = ["lib.hip", "attention.hip", "norm.hip", "elementwise.hip", "audio.hip"]
# gemm.hip instantiates ~270 kernels: compiled as four translation units (one per A-loader mode + the entry point)
GEMM_UNITS = 4
HEADERS = [os.path.join(CSRC, "avsd_common.h"), os.path.join(HERE, "..", "include", "avsd.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


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


def build(verbose: bool = False, force: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            jobs.append([hipcc, *FLAGS, "-c", s, "-o", o])
    gsrc = os.path.join(CSRC, "gemm.hip")
    for u in range(GEMM_UNITS):
        o = os.path.join(OBJ, f"gemm_tu{u}.o")
        objs.append(o)
        if force or _stale(o, [gsrc] + HEADERS):
            jobs.insert(0, [hipcc, *FLAGS, f"-DAVSD_GEMM_TU={u}", "-c", gsrc, "-o", o])     # longest jobs first

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
