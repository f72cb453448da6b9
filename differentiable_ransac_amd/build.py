"""Builds libdransac.so (HIP, gfx950 only) in-tree with hipcc.

    python -m differentiable_ransac_amd.build [--force]

One object per .hip file (compiled in parallel), linked into
differentiable_ransac_amd/libdransac.so.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libdransac.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wall", "-Wno-unused-function"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "dransac.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ, src[:-4] + ".o")
    spath = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(spath)
            and os.path.getmtime(obj) > _deps_mtime()):
        return obj
    cmd = [HIPCC, *FLAGS, "-c", spath, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build_library(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv, verbose=True)
