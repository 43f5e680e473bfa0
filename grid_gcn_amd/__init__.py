"""grid_gcn_amd: the Grid-GCN hot path (CAGQ index ops, neighbour gather, GridConv) on MI355X.
Importing the package registers torch.ops.gridgcn.* (torch_ops.py); the HIP library itself is
loaded on first use (_lib.load) and there is no CPU fallback."""
from . import torch_ops  # noqa: F401
