"""Regenerate tests/golden/gridconv_*.npz: GridConv layer outputs (train and eval BatchNorm) of
the INDEPENDENT float64 restatement oracle/gridconv_ref.py on the seeded cases of
tests/gridconv_cases.py.

    python tests/golden/make_gridconv_golden.py

The reference (MXNet) can be neither built nor imported here, so these are not MXNet
outputs: they pin the product's float path (stock-op modules AND HIP kernels) against a second,
separately written restatement of the reference graphs.  Stored: float32 of the float64 result,
centre rows `rows` of the case only (fixtures stay a few hundred KB).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import gridconv_cases as gc  # noqa: E402

if __name__ == "__main__":
    for name, build in gc.CASES.items():
        case = build()
        d = {}
        for mode, train in (("train", True), ("eval", False)):
            out = gc.reference_output(case, train)
            d[mode] = out[:, case["rows"], :].astype(np.float32)
            d[mode + "_absmax"] = np.array(np.abs(out).max())
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, d["train"].shape, "absmax train %.3f eval %.3f" % (d["train_absmax"], d["eval_absmax"]),
              "%.0f KB" % (os.path.getsize(os.path.join(HERE, name + ".npz")) / 1024))
