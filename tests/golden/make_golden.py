"""Regenerate tests/golden/*.npz from the CPU oracle (oracle/gridgcn_oracle.c).

    python tests/golden/make_golden.py

The reference holds no golden vectors and can be neither built nor imported in
this image (F6), so these vectors pin the ORACLE (against drift) and the HIP kernels (against
the oracle) -- they are not reference outputs.  Each fixture stores, per output tensor, its
shape, dtype, SHA-256 and the leading 4096 elements; inputs are regenerated from seeds by
tests/cases.py.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

from oracle import oracle as orc  # noqa: E402
import cases  # noqa: E402

HEAD = 4096


def summarise(outs):
    d = {}
    for j, a in enumerate(outs):
        a = np.ascontiguousarray(a)
        d["sha%d" % j] = np.frombuffer(hashlib.sha256(a.tobytes()).digest(), np.uint8)
        d["shape%d" % j] = np.array(a.shape, np.int64)
        d["head%d" % j] = a.reshape(-1)[:HEAD].copy()
    d["n"] = np.array(len(outs))
    return d


def all_cases():
    out = []
    for name, build in cases.gridify_cases(orc.gridify):
        out.append((name, build, lambda a, kw: orc.gridify(*a, **kw)))
    for name, build in cases.gridify_knn_cases(orc.gridify):
        out.append((name, build, lambda a, kw: orc.gridify_knn(*a, **kw)))
    for name, build in cases.gridify_up_cases(orc.gridify):
        out.append((name, build, lambda a, kw: orc.gridify_up(*a, **kw)))
    for name, build in cases.gridify_variant_cases():
        if name.startswith("occaware"):
            out.append((name, build, lambda a, kw: orc.gridify_occaware(*a, **kw)))
        else:
            out.append((name, build, lambda a, kw: orc.gridify_fast_rand(*a, **kw)))
    for name, build in cases.knn_cases():
        out.append((name, build, lambda a, kw: (orc.ball_knn(*a, **kw),)))
        out.append((name.replace("ball_knn", "knn"), build,
                    lambda a, kw: (orc.knn(*a, k=kw["k"]),)))
    return out


if __name__ == "__main__":
    for name, build, run in all_cases():
        args, kw = build()
        outs = run(args, kw)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **summarise(outs))
        print(name, [tuple(o.shape) for o in outs])
