"""Build libdsge_hip.so (hipcc, gfx950 only) in-tree.

    python -m geconpy_amd.build            # incremental
    python -m geconpy_amd.build --force

Every ``csrc/*.hip`` file is one translation unit (the C ABI + one launcher file per kernel
family); they are compiled to objects in parallel and linked into one shared library.  hipcc
cross-compiles without a GPU, so this runs in the build container; the resulting
``geconpy_amd/libdsge_hip.so`` is git-ignored but travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "_obj")
LIB = os.path.join(PKG, "libdsge_hip.so")
SOURCES = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp")) + [os.path.join("..", "..", "include", "dsge_hip.h")]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libdsge_hip.so")


def _includes(path, seen=None):
    """Transitive closure of the quoted #include files of a source (paths relative to its directory)."""
    seen = set() if seen is None else seen
    with open(path) as fh:
        for line in fh:
            line = line.strip()
            if line.startswith('#include "'):
                inc = os.path.normpath(os.path.join(os.path.dirname(path), line.split('"')[1]))
                if inc not in seen and os.path.exists(inc):
                    seen.add(inc)
                    _includes(inc, seen)
    return seen


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


STAMP = LIB + ".sha256"  # travels with the library (git-ignored, not gpurun-ignored)


def sources_digest():
    """Content hash of every source the library is built from (csrc/*.hip, csrc/*.hpp, include/dsge_hip.h) and of the
    compiler flags: what `libdsge_hip.so` is checked against, so that a prebuilt library travelling with a snapshot can
    never be older than the sources next to it, whatever the file times say."""
    import hashlib

    h = hashlib.sha256(" ".join(CFLAGS).encode())
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != sources_digest()


def build_library(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(OBJ, src[:-4] + ".o")
        path = os.path.join(CSRC, src)
        if force or _stale(obj, [path, *_includes(path)]):
            cmd = [hipcc, *CFLAGS, "-c", "-o", obj, path]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True, cwd=CSRC)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    with open(STAMP, "w") as fh:
        fh.write(sources_digest())
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(LIB)
