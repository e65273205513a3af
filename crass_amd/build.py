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
SOURCES = ["kernels.hip", "dmerge.hip", "consensus.hip", "engine.cpp", "merge.cpp", "ingest.cpp", "consensus.cpp", "group.cpp", "graph.cpp", "sdma.cpp", "pgzip.cpp"]
DEPS = SOURCES + ["engine_internal.h", "devmem.h", "consensus_internal.h", "merge.h", os.path.join("..", "..", "include", "crass_hip.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X engine cannot be built")


OBJ = os.path.join(CSRC, "_obj")
HEADERS = ["engine_internal.h", "devmem.h", "consensus_internal.h", "merge.h", os.path.join("..", "..", "include", "crass_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall", "-Wno-unused-function"]
LIBS = ["-lz", "-lpthread", "-ldl"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build():
    return _stale(LIB, [os.path.join(CSRC, d) for d in DEPS])


def build(force=False, verbose=False):
    """One object per source (compiled in parallel, only the stale ones), then one link; the objects live in
    csrc/_obj (git-ignored).  Every kernel is compiled for gfx950 only."""
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    objs = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc] + FLAGS + ["-x", "hip", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            jobs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = []
    for s, p in jobs:
        out, _ = p.communicate()
        if out.strip():
            print(out, file=sys.stderr, end="")
        if p.returncode:
            failed.append(s)
    if failed:
        raise RuntimeError("hipcc failed for: " + ", ".join(failed))
    tmp = LIB + ".tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs + LIBS
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    from . import vgpr_guard
    try:
        vgpr_guard.check(tmp)      # no kernel may use the last VGPR of its allocation (engine_internal.h, CRASS_VGPR_FLOOR)
    except Exception:
        os.replace(tmp, LIB + ".rejected")      # a library that failed the check must not be loadable by accident
        raise
    os.replace(tmp, LIB)
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
