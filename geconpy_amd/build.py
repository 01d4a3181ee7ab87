"""Build libdsge_hip.so (hipcc, gfx950 only) in-tree.

    python -m geconpy_amd.build            # incremental
    python -m geconpy_amd.build --force

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
``geconpy_amd/libdsge_hip.so`` is git-ignored but travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libdsge_hip.so")
SOURCES = ["dsge_api.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp")) + [os.path.join("..", "..", "include", "dsge_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libdsge_hip.so")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), *FLAGS, "-o", LIB, *[os.path.join(CSRC, s) for s in SOURCES]]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(LIB)
