"""Compile libgridgcn_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m grid_gcn_amd.build [--force]

One object per source (compiled in parallel, rebuilt only when the source or a header changed),
then one link.  The .so is written in-tree (grid_gcn_amd/lib/) so that it travels with the source
snapshot to the GPU box; it is git-ignored.
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libgridgcn_hip.so")
SOURCES = ["gridgcn_index.hip", "gridgcn_index_legacy.hip", "gridgcn_query.hip",
           "gridgcn_query_knn.hip", "gridgcn_knn.hip", "gridgcn_conv.hip", "gridgcn_train.hip",
           "gridgcn_direct.hip", "gridgcn_bwdfused.hip", "gridgcn_attbwd.hip", "gridgcn_attbwd_nz.hip", "gridgcn_attfwd.hip", "gridgcn_atteval.hip",
           "gridgcn_pairmax.hip", "gridgcn_head.hip", "gridgcn_scatter.hip",
           "gridgcn_edgelin.hip", "gridgcn_ballgrid.hip", "gridgcn_clsblock.hip", "gridgcn_cas.hip", "gridgcn_fastrand.hip",
           "gridgcn_gemm.hip", "gridgcn_optim.hip",
           "gridgcn_capi.hip"]
# -ffp-contract=off: the parity contract is "fp32, IEEE, no FMA contraction" (SURVEY App. A);
# the MFMA/FMA use inside the GridConv kernels is explicit (intrinsics), never compiler-made.
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall",
          "-Wno-unused-function", "-Wno-unused-variable"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (needed to build libgridgcn_hip.so)")


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "gridgcn.h"))
    return max(os.path.getmtime(h) for h in hs if os.path.exists(h))


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _obj(src):
    return os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")


def _stale(src, hm):
    o = _obj(src)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    return os.path.getmtime(os.path.join(CSRC, src)) > t or hm > t


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    hm = _headers_mtime()
    return hm > t or any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in _sources())


def build(force=False, verbose=False, prof=False):
    """prof=True: the -DGG_PROF variant (phase stamps inside the index/query kernels, read by
    tools/prof_phases.py) as lib/libgridgcn_hip_prof.so; never loaded unless GG_HIP_LIB says so."""
    global OBJDIR, LIB
    if prof:
        saved = (OBJDIR, LIB)
        OBJDIR, LIB = os.path.join(LIBDIR, "obj_prof"), os.path.join(LIBDIR, "libgridgcn_hip_prof.so")
        CFLAGS.append("-DGG_PROF")
        try:
            return build(force=force, verbose=verbose)
        finally:
            CFLAGS.remove("-DGG_PROF")
            OBJDIR, LIB = saved
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    cc = _hipcc()
    hm = _headers_mtime()
    srcs = _sources()
    todo = [s for s in srcs if force or _stale(s, hm)]

    def compile_one(s):
        cmd = [cc] + CFLAGS + ["-c", os.path.join(CSRC, s), "-o", _obj(s) + ".tmp"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        os.replace(_obj(s) + ".tmp", _obj(s))

    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, os.cpu_count() or 1)) as ex:
        list(ex.map(compile_one, todo))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj(s) for s in srcs] + \
          ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, prof="--prof" in sys.argv))
