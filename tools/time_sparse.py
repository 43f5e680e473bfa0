"""Device time of gridgcn_edge_lin0_backward_sparse at the cfg4 up2 shape (8 clouds, 1024 source
rows, 81920 centres x 5 neighbours, 128 channels) on random inputs; GG_HIP_LIB selects a variant
library (tools/micro/sparsevar.sh)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grid_gcn_amd import _lib

lib = _lib.load()
dev = "cuda"
B, N, O, P, C = 8, 1024, 81920, 5, 128
torch.manual_seed(0)
E = B * O * P
nebidx = torch.randint(0, N, (B, O, P), device=dev, dtype=torch.int32)
att16 = torch.randn(E, 16, device=dev)
amax = torch.randint(0, P, (B * O, C), device=dev, dtype=torch.uint8)
gval = torch.randn(B * O, C, device=dev)
zsel = torch.randn(B * O, C, device=dev)
Ysrc = torch.randn(B * N, C, device=dev)
wgb = torch.randn(4, C, device=dev)
vec = [torch.rand(C, device=dev) + 0.5 for _ in range(6)]
dYsrc = torch.empty(B * N, C, device=dev)
Gsum = torch.empty(B * N, 4, device=dev)
acc64 = torch.zeros(3 * C + 12, dtype=torch.float64, device=dev)
nb = ctypes.c_size_t(0)
lib.gridgcn_edge_lin0_backward_sparse_workspace_bytes(B, N, C, ctypes.byref(nb))
ws = torch.empty(nb.value, dtype=torch.uint8, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def call():
    rc = lib.gridgcn_edge_lin0_backward_sparse(
        p(nebidx), p(att16), p(amax), p(gval), p(zsel), p(Ysrc), p(wgb), p(wgb[3]), p(vec[0]), p(vec[1]),
        p(vec[2]), p(vec[3]), p(vec[4]), p(vec[5]), B, N, O, P, C, p(dYsrc), p(Gsum), p(acc64),
        p(acc64[3 * C:]), p(ws), nb.value, st)
    assert rc == 0, rc


for _ in range(5):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    call()
e1.record()
torch.cuda.synchronize()
print("%s: sparse + finish %.1f us per call" % (os.environ.get("GG_HIP_LIB", "default"), e0.elapsed_time(e1) * 50))
