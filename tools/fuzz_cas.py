"""Seed sweep of Gridify_occaware against the one-by-one restatement (oracle) on the stress shapes of
tests/test_cas.py: python tools/fuzz_cas.py [nseeds]   (GPU; the oracle runs on the host)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

import test_cas as t
from grid_gcn_amd import ops, synth
from oracle import oracle as orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
bad = tot = 0
for case in t.CASES:
    name, cfg, layer, N, kind, over = case
    over = dict(over)
    beta = over.pop("beta", 1.0)
    over.pop("ragged", None)
    for seed in range(n):
        data, npn = synth.make_batch(2, N, kind)
        rng = np.random.default_rng(seed)
        data = data[:, rng.permutation(N)]                      # another order of first appearance per seed
        kw = t._kw(cfg, layer, seed=seed * 7919 + 1, **over)
        want = orc.gridify_occaware(data, npn, beta=beta, **kw)
        got = ops.Gridify_occaware(torch.from_numpy(np.ascontiguousarray(data)).cuda(), torch.from_numpy(npn).cuda(),
                                   beta=beta, **kw)
        ok = all(np.array_equal(w, g.cpu().numpy()) for w, g in zip(want, got))
        tot += 1
        bad += 0 if ok else 1
        if not ok:
            print("MISMATCH", name, seed)
print("%d / %d runs bit-exact" % (tot - bad, tot))
sys.exit(1 if bad else 0)
