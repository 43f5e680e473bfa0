"""Device time of one Gridify call (layer 0) for every setting of the index build's two plan
options (GRIDGCN_OPT_INDEX_SLAB_SHIFT, GRIDGCN_OPT_INDEX_CHUNK):
    python tools/sweep_index.py [--cfg seg80k|synth200k|seg8k|cls]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grid_gcn_amd import _lib, ops, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="seg80k")
ap.add_argument("--B", type=int, default=8)
a = ap.parse_args()
cfg = {"seg80k": synth.SEG_SCANNET_81920, "seg8k": synth.SEG_SCANNET_8192,
       "cls": synth.CLS_MODELNET40, "synth200k": synth.SYNTH_200K}[a.cfg]
lib = _lib.load()
data, npn = synth.make_batch(a.B, cfg["num_points"], "planes" if a.cfg != "cls" else "ball")
d, n = torch.from_numpy(data).cuda(), torch.from_numpy(npn).cuda()
kw = synth.gridify_kwargs(cfg, 0)
for ch in (0, 1024, 2048, 4096):
    for sh in (-2, -1, 0, 1, 2):
        _lib.check(lib.gridgcn_set_option(_lib.OPT_INDEX_CHUNK, ch), "opt")
        _lib.check(lib.gridgcn_set_option(_lib.OPT_INDEX_SLAB_SHIFT, sh), "opt")
        ms, _ = ops.gridify_timed(d, n, 50, **kw)
        print("%s chunk %4d slab shift %+d: %.1f us" % (a.cfg, ch, sh, ms * 1e3), flush=True)
lib.gridgcn_set_option(_lib.OPT_INDEX_CHUNK, 0)
lib.gridgcn_set_option(_lib.OPT_INDEX_SLAB_SHIFT, 0)
