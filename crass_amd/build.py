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
SOURCES = ["kernels.hip", "dmerge.hip", "consensus.hip", "engine.cpp", "merge.cpp", "ingest.cpp", "consensus.cpp"]
DEPS = SOURCES + ["engine_internal.h", "consensus_internal.h", "merge.h", os.path.join("..", "..", "include", "crass_hip.h")]


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
    cmd += ["-lz", "-lpthread", "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    from . import vgpr_guard
    try:
        vgpr_guard.check(LIB)      # no kernel may use the last VGPR of its allocation (engine_internal.h, CRASS_VGPR_FLOOR)
    except Exception:
        os.replace(LIB, LIB + ".rejected")      # a library that failed the check must not be loadable by accident
        raise
    return LIB


ADAPTER_DIR = os.path.join(CSRC, "adapter")
CLI = os.path.join(HERE, "crass-hip")
ADAPTER_LIB = os.path.join(HERE, "libcrass_adapter.so")


def build_adapter(force=False, verbose=False):
    """C++ host adapter with the reference's seam (searchFile/createNonRedundantSet/findSingletons)
    + the `crass-hip` command line.  Plain g++; links against libcrass_hip.so."""
    build(force=force, verbose=verbose)
    srcs = [os.path.join(ADAPTER_DIR, "crass_adapter.cpp")]
    deps = srcs + [os.path.join(ADAPTER_DIR, "crass_adapter.h"), os.path.join(ADAPTER_DIR, "crass_hip_cli.cpp"), LIB]
    if not force and os.path.exists(CLI) and os.path.exists(ADAPTER_LIB) and \
            all(os.path.getmtime(d) <= min(os.path.getmtime(CLI), os.path.getmtime(ADAPTER_LIB)) for d in deps):
        return CLI
    cxx = shutil.which("g++") or "g++"
    common = [cxx, "-O2", "-std=c++17", "-Wall", "-fPIC", "-pthread"]
    rpath = ["-L" + HERE, "-lcrass_hip", "-Wl,-rpath," + HERE, "-Wl,-rpath,$ORIGIN"]
    for cmd in (common + ["-shared", "-o", ADAPTER_LIB] + srcs + rpath,
                common + ["-o", CLI, os.path.join(ADAPTER_DIR, "crass_hip_cli.cpp")] + srcs + rpath):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return CLI


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_adapter(force="--force" in sys.argv, verbose=True))
