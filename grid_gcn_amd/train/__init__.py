"""Host side of the training / evaluation path on the hand-written kernels, by block (round 5: was one
2 570-line module; the re-export shim `grid_gcn_amd/train_ops.py` that bridged the split was deleted in round 6 -- callers
import the block they use: `from grid_gcn_amd.train import edge`, `from grid_gcn_amd.train.options import OPT`).

    options    the path switches: one object, OPT
    common     operand packing cache, conv + BatchNorm + ReLU chains (forward / backward), small GEMMs, glue
    mlp        per-point stacks (centre / update MLPs, fc1), wide layers
    edge       segmentation GridConv edge block (source-side first conv, attention chain, product + max)
    cls        classification GridConv edge block
    head       class scores, fc1 + Dropout + fc2, softmax cross-entropy
    evalpath   evaluation through the same kernels with running statistics; caches of folded constants
    timers     micro-benchmarks of single library calls (bench.py's roofline lines)
"""
