"""Do two independent chains of SMALL launches overlap when they are captured as two branches of one hipGraph?
(0.8 ms of a cfg4 step runs on fewer than 256 workgroups; the dW + reduce launches of the coarse layers
do not feed the dX chain.)  Chains of the library's own small kernels (gridgcn_linear_bwd on 2048-row layers: dX,
dW, reduce = three launches on 16-64 workgroups), one stream against fork / join on two streams, replayed."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grid_gcn_amd import _lib
from grid_gcn_amd.train import common as tcommon

dev = "cuda:0"
E, cin, C = 2048, 128, 128


def make_call(seed):
    # one gridgcn_linear_bwd call on its own tensors (dense gradient), as ttimers.time_linear_bwd builds it
    import types
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
    from grid_gcn_amd.ops import _ptr
    Z, X, dY = rnd(E, C), rnd(E, cin), rnd(E, C)
    v = [rnd(C).abs() + 0.5, rnd(C) * 0.1, rnd(C) * 0.1, rnd(C).abs() + 0.5, rnd(C) * 1e-3, rnd(C) * 1e-3]
    Wt = rnd(C, cin)
    Wb, Wg = tcommon.pack_tiles(Wt), tcommon.pack_groups(Wt)
    Wdx = torch.empty(C * 32 * 8, device=dev)
    lib.gridgcn_pack_linear(_ptr(Wt), None, C, cin, 0, cin, cin, None, None, None, None, None, _ptr(Wdx),
                            torch.cuda.current_stream().cuda_stream)
    dX, dW = torch.empty(E, cin, device=dev), torch.empty(C, cin, device=dev)
    nb = ctypes.c_size_t(0)
    lib.gridgcn_linear_bwd_workspace_bytes(E, cin, C, ctypes.byref(nb))
    ws = torch.empty(nb.value, dtype=torch.uint8, device=dev)
    keep = (Z, X, dY, v, Wb, Wg, Wdx, dX, dW, ws)

    def call():
        rc = lib.gridgcn_linear_bwd(_ptr(dY), _ptr(Z), _ptr(v[0]), _ptr(v[1]), _ptr(v[2]), _ptr(v[3]), _ptr(v[4]),
                                    _ptr(v[5]), _ptr(X), None, None, None, None, _ptr(Wb), _ptr(Wg), _ptr(Wdx), cin, E, C,
                                    cin, cin, 0, C, _ptr(dX), _ptr(dW), None, None, None, 1, _ptr(ws), nb.value,
                                    torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    call.keep = keep
    return call


NCH = 10
A = [make_call(i) for i in range(NCH)]
B = [make_call(100 + i) for i in range(NCH)]
torch.cuda.synchronize()


def one_stream():
    for a, b in zip(A, B):
        a(); b()


def two_streams():
    s2 = two_streams.s2
    cur = torch.cuda.current_stream()
    s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        for b in B:
            b()
    for a in A:
        a()
    cur.wait_stream(s2)


two_streams.s2 = torch.cuda.Stream()
for name, fn in (("one stream", one_stream), ("two branches", two_streams)):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            fn()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print("%-14s %d + %d calls of linear_bwd(E=%d): %.1f us per replay" % (name, NCH, NCH, E, e0.elapsed_time(e1) * 20))
