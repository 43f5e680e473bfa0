"""Host-side guards (CPU): option values of the reference's sub_g_update that this package does not
restate are refused, never silently ignored; the C library reads nothing from the environment; the
device-side seed increment is only handed to the index operators where a fresh draw per call is the
contract."""
import glob
import os

import pytest
import torch

from grid_gcn_amd.gridconv import SubGUpdate
from grid_gcn_amd.model_cls import SubGUpdateCls

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("kw", [dict(attfdim=5), dict(attfdim=11), dict(attfdim=12), dict(attfdim=4),
                                dict(localfdim=4), dict(localfdim=12), dict(elevation=[16]),
                                dict(aggtype="agg_gcn"), dict(up_center_inte="add"),
                                dict(att_full="last"), dict(att_full="next"),
                                dict(pool_type="avg_pooling"), dict(cntxt_mlp=[64, 64])])
def test_seg_sub_g_update_refuses_unshipped_branches(kw):
    """segmentation/models/gcn_module_g_att.py:196-282: branches no shipped yaml uses."""
    SubGUpdate(64, [64, 128], localfdim=3)                       # the shipped form builds
    with pytest.raises(NotImplementedError) as e:
        SubGUpdate(64, [64, 128], **dict(dict(localfdim=3), **kw))
    assert "gcn_module_g_att.py" in str(e.value)


@pytest.mark.parametrize("kw", [dict(attfdim=10), dict(attfdim=5), dict(localfdim=5),
                                dict(elevation=[8]), dict(aggtype="agg_gcn"),
                                dict(up_center_inte="add"), dict(att_full="last"), dict(att_full=""),
                                dict(cntxt_mlp=[64])])
def test_cls_sub_g_update_refuses_unshipped_branches(kw):
    """classification/models/gcn_module_g.py:116-209 with configs.yaml:47-63 as the shipped form."""
    SubGUpdateCls(128, [128, 128, 256], [128, 256, 256])
    with pytest.raises(NotImplementedError):
        SubGUpdateCls(128, [128, 128, 256], [128, 256, 256], **kw)


def test_c_library_reads_no_environment():
    for f in glob.glob(os.path.join(ROOT, "grid_gcn_amd", "csrc", "*")):
        assert "getenv" not in open(f).read(), f


def test_seed_dev_only_in_training_without_fixed_seed():
    """after graph.GraphedTrainStep set net.seed_dev, evaluation (and fixed_seed nets)
    must still sample with `seed` itself."""
    from grid_gcn_amd import model, model_cls, model_synth
    from oracle.torch_index_ops import OracleIndexOps
    net = model.GGCNSeg(model.SEG_8192, index_ops=OracleIndexOps, seed=7)
    net.seed_dev = torch.zeros(1, dtype=torch.int64)
    assert net.train()._seed_dev() is net.seed_dev
    assert net.eval()._seed_dev() is None
    fixed = model.GGCNSeg(model.SEG_8192, index_ops=OracleIndexOps, seed=7, fixed_seed=True)
    fixed.seed_dev = torch.zeros(1, dtype=torch.int64)
    assert fixed.train()._seed_dev() is None

    class Spy(model.HipIndexOps):
        calls = []

        @staticmethod
        def Gridify(data, num, **kw):
            Spy.calls.append(kw.get("seed_dev"))
            raise StopIteration

    for mk in (lambda **k: model.GGCNSeg(model.SEG_8192, index_ops=Spy, **k),
               lambda **k: model_cls.GGCNCls(index_ops=Spy, **k),
               lambda **k: model_synth.GGCNSynth(index_ops=Spy, **k)):
        for train, fixed_seed, want in ((True, False, True), (False, False, False),
                                        (True, True, False)):
            n = mk(seed=3, fixed_seed=fixed_seed)
            n.seed_dev = torch.zeros(1, dtype=torch.int64)
            n.train(train)
            Spy.calls.clear()
            with pytest.raises(StopIteration):
                n(torch.zeros(1, 64, 3), torch.full((1, 1), 64, dtype=torch.int32))
            assert (Spy.calls[0] is not None) == want, (type(n).__name__, train, fixed_seed)
