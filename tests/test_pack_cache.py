"""tcommon._PackCache bookkeeping (CPU tier: the pack launch itself is replaced by a counter; the
descriptor fill is host code of the C library).  ADVICE r3: (1) a Parameter that got new storage must
not leave its old entry -- dead pointers -- in the cache or in a device table; (2) a re-pack between
a forward and its backward must not pass silently."""
import pytest
import torch

from grid_gcn_amd import _lib
from grid_gcn_amd.train import common as tcommon


@pytest.fixture
def cache(monkeypatch):
    packs = []
    monkeypatch.setattr(tcommon._PackCache, "_pack_one",
                        staticmethod(lambda lib, W, b, *a: packs.append((W.data_ptr(), b.data_ptr()))))
    c = tcommon._PackCache()
    c.packs = packs
    return c


def _layer(cout=32, cin=16):
    W = torch.nn.Parameter(torch.randn(cout, cin))
    b = torch.nn.Parameter(torch.randn(cout))
    sizes = tcommon.packed_sizes(cout, cin)
    return W, b, (sizes[2], sizes[1], sizes[3], cin * sizes[1], cout * 32)


def _get(c, W, b, sizes):
    cout, cin = W.shape
    return c.get(_lib.load(), W, b, cout, cin, 0, cin, cin, True, sizes, None)


def test_pack_cache_drops_entries_of_moved_storage(cache):
    W, b, sizes = _layer()
    _get(cache, W, b, sizes)
    assert len(cache.entries) == 1
    old = next(iter(cache.entries))
    # the same Parameter objects, new storage (what net.to(dev) / param.data = ... do)
    W.data = W.data.clone()
    b.data = b.data.clone()
    _get(cache, W, b, sizes)
    assert len(cache.entries) == 1 and old not in cache.entries
    (k, e), = cache.entries.items()
    assert (e["desc"].W, e["desc"].b) == (W.data_ptr(), b.data_ptr()) == k[8:10]
    assert cache.packs[-1] == (W.data_ptr(), b.data_ptr())


def test_pack_cache_table_never_lists_dead_pointers(cache):
    """prepack's key list is what becomes the device-side descriptor table: after a move it holds the
    live entries only (the GPU part of prepack -- the table upload and the batch launch -- is not run
    here)."""
    lin = torch.nn.Linear(16, 32)
    lin2 = torch.nn.Linear(16, 32)
    mod = torch.nn.ModuleList([lin, lin2])
    sizes = _layer()[2]
    for l in (lin, lin2):
        _get(cache, l.weight, l.bias, sizes)
    lin.weight.data = lin.weight.data.clone()         # moved, and not looked up again yet
    ids = {id(p) for p in mod.parameters()}
    assert sum(not cache._live(k) for k in cache.entries) == 1
    for k in [k for k in cache.entries if k[0] in ids and not cache._live(k)]:
        cache._drop(k)
    assert all(cache._live(k) for k in cache.entries) and len(cache.entries) == 1


def test_pack_cache_repack_between_forward_and_backward_is_an_error(cache):
    """The layouts live in shared persistent buffers and save_for_backward keeps views of them; a
    second forward (= a re-pack) before the first one's backward used to change that backward's
    operands silently.  Every re-pack now moves the buffer's version counter, so autograd refuses."""
    W, b, sizes = _layer()

    class UsesPack(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            bufs = _get(cache, W, b, sizes)
            ctx.save_for_backward(bufs[2])
            return x * 2

        @staticmethod
        def backward(ctx, g):
            ctx.saved_tensors
            return g * 2

    x = torch.ones(3, requires_grad=True)
    UsesPack.apply(x).sum().backward()                # forward -> backward: fine, every step
    UsesPack.apply(x).sum().backward()
    y1 = UsesPack.apply(x)
    UsesPack.apply(x)                                 # re-pack while y1's backward is outstanding
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        y1.sum().backward()
