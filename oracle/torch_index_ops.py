"""TEST INFRASTRUCTURE: index-operator provider backed by the CPU oracle, with the same
interface as grid_gcn_amd.model.HipIndexOps.  Lets the GridConv/model float path run on CPU
(tests, bench.py's cpu_baseline leg).  Never imported by the product package."""
import numpy as np
import torch

from . import oracle as orc


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


class OracleIndexOps:
    @staticmethod
    def Gridify(data, actual_numpoints, **kw):
        return tuple(_t(x) for x in orc.gridify(data.detach().cpu().numpy(),
                                                actual_numpoints.cpu().numpy(), **kw))

    @staticmethod
    def GridifyUp(down, up, dn, un, **kw):
        return tuple(_t(x) for x in orc.gridify_up(down.detach().cpu().numpy(),
                                                   up.detach().cpu().numpy(), dn.cpu().numpy(),
                                                   un.cpu().numpy(), **kw))

    @staticmethod
    def BallKNN(unknown, known, downnum, upnum, *, k=3, radius=0.1):
        return _t(orc.ball_knn(unknown.detach().cpu().numpy(), known.detach().cpu().numpy(),
                               downnum.cpu().numpy(), upnum.cpu().numpy(), k=k, radius=radius))

    @staticmethod
    def batch_take_g(data, index):
        """differentiable torch restatement of utils/ops.py:78-93 (flat take, mode='clip')."""
        B, N, C = data.shape
        flat = index.long() + (torch.arange(B, device=index.device) * N).view(
            B, *([1] * (index.dim() - 1)))
        flat = flat.clamp(0, B * N - 1)
        return data.reshape(B * N, C)[flat]


class OracleIndexOpsKNN(OracleIndexOps):
    """centre neighbours from the S0 restatement of GridifyKNN (gridifyknn.cu:115-204, 231-332)"""
    @staticmethod
    def Gridify(data, actual_numpoints, **kw):
        return tuple(_t(x) for x in orc.gridify_knn(data.detach().cpu().numpy(),
                                                    actual_numpoints.cpu().numpy(), **kw))
