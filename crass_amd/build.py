"""Builds libcrass_hip.so (HIP kernels for gfx950 + C-ABI host engine) in-tree with hipcc.

The .so is git-ignored but travels to the GPU box with the repo snapshot.  hipcc
cross-compiles gfx950 code objects without a GPU."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcrass_hip.so")
SOURCES = ["kernels.hip", "engine.cpp", "merge.cpp", "ingest.cpp"]
DEPS = SOURCES + ["engine_internal.h", "merge.h", os.path.join("..", "..", "include", "crass_hip.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X engine cannot be built")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", "-o", LIB]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-lz", "-lpthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
