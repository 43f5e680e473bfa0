"""Compile libgridgcn_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m grid_gcn_amd.build [--force]

The .so is written in-tree (grid_gcn_amd/lib/) so that it travels with the source snapshot to
the GPU box; it is git-ignored.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libgridgcn_hip.so")
SOURCES = ["gridgcn_index.hip", "gridgcn_query.hip", "gridgcn_query_knn.hip", "gridgcn_knn.hip",
           "gridgcn_conv.hip", "gridgcn_train.hip", "gridgcn_direct.hip", "gridgcn_attbwd.hip", "gridgcn_atteval.hip", "gridgcn_pairmax.hip", "gridgcn_head.hip", "gridgcn_scatter.hip", "gridgcn_edgelin.hip", "gridgcn_ballgrid.hip", "gridgcn_capi.hip"]
# -ffp-contract=off: the parity contract is "fp32, IEEE, no FMA contraction" (SURVEY App. A);
# the MFMA/FMA use inside the GridConv kernels is explicit (intrinsics), never compiler-made.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (needed to build libgridgcn_hip.so)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(HERE, "..", "include", "gridgcn.h"))
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [_hipcc()] + FLAGS + srcs + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
