import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from grid_gcn_amd import ops
from grid_gcn_amd.gridconv import SubGUpdate
DEV = "cuda:0"
seed = int(sys.argv[1])
rng = np.random.default_rng(3000 + seed)
torch.manual_seed(seed)
cin = int(rng.choice([0, 4, 16, 33, 64, 128, 260])); L = int(rng.integers(1, 4))
dims = [int(rng.choice([16, 32, 64, 128, 256])) for _ in range(L)]
if rng.random() < 0.2: dims[-1] = 512
lfd = int(rng.choice([0, 3])); P = int(rng.choice([1, 5, 8, 32, 64])); O = int(rng.choice([7, 64, 300]))
B, Nsrc = int(rng.integers(1, 4)), int(rng.choice([50, 400]))
up = cin > 0 and rng.random() < 0.4
import json
ov = json.loads(os.environ.get('OV', '{}'))
cin = ov.get('cin', cin); dims = ov.get('dims', dims); lfd = ov.get('lfd', lfd); P = ov.get('P', P); O = ov.get('O', O); B = ov.get('B', B); Nsrc = ov.get('Nsrc', Nsrc); up = ov.get('up', up)
kwargs = dict(center_in=4 + 32, center_dim=[64], out_dim=[64]) if up else {}
ref = SubGUpdate(cin, dims, localfdim=lfd, **kwargs).train()
for m in ref.modules():
    if isinstance(m, torch.nn.BatchNorm1d):
        m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.3)
gen = torch.Generator().manual_seed(seed)
src = torch.rand(B, Nsrc, 4 + cin, generator=gen) * 2 - 1
nebidx = torch.randint(-1, Nsrc, (B, O, P), generator=gen, dtype=torch.int32)
cent = torch.rand(B, O, 4, generator=gen) * 2 - 1
cof = (torch.rand(B, O, 36, generator=gen) * 2 - 1) if up else None
msk = (torch.rand(B, O, generator=gen) > 0.2).float() if rng.random() < 0.5 else None
cot = torch.randn(B, O, ref.out_channels, generator=gen)
print("cin", cin, "dims", dims, "lfd", lfd, "P", P, "O", O, "B", B, "Nsrc", Nsrc, "up", up, "msk", msk is not None)
def run(dev, dtype, kernel):
    m = copy.deepcopy(ref).to(dev).to(dtype).train(); m.mfma_train = kernel
    s = src.to(dev).to(dtype).requires_grad_(cin > 0)
    c = cent.to(dev).to(dtype); ix = nebidx.to(dev)
    cf = None if cof is None else cof.to(dev).to(dtype); mk = None if msk is None else msk.to(dev).to(dtype)
    if kernel:
        y = m.forward_src(c, s, ix, mk, center_ori_feats=cf)
    else:
        Bn, N, C = s.shape
        flat = (ix.long() + (torch.arange(Bn, device=dev) * N).view(Bn, 1, 1)).clamp(0, Bn * N - 1)
        y = m(c[..., 0:3], s.reshape(Bn * N, C)[flat], mk, center_ori_feats=cf)
    g = cot.to(dev).to(dtype)
    y.backward(g)
    out = {"y": y.detach().double().cpu()}
    if cin: out["src"] = s.grad[..., 4:].double().cpu()
    for n_, p in m.named_parameters():
        if p.grad is not None and not n_.endswith("lin.bias"): out[n_] = p.grad.double().cpu()
    return out
r64 = run("cpu", torch.float64, False); rs = run(DEV, torch.float32, False); rk = run(DEV, torch.float32, True)
for k in r64:
    s_ = max(float(r64[k].abs().max()), 1e-30)
    es, ek = float((rs[k] - r64[k]).abs().max()) / s_, float((rk[k] - r64[k]).abs().max()) / s_
    print("%-28s stock32 %.2e  kernels %.2e%s" % (k, es, ek, "  <<<" if ek > max(3 * es, 1e-4) else ("  (stock off)" if es > 1e-4 else "")))
# near-ties of the neighbour max in float64
m = copy.deepcopy(ref).double().train()
s = src.double(); Bn, N, C = s.shape
flat = (nebidx.long() + (torch.arange(Bn) * N).view(Bn, 1, 1)).clamp(0, Bn * N - 1)
nf, att_vec = m.edge_inputs(s.reshape(Bn * N, C)[flat], cent.double()[..., 0:3])
pair = m.att2(m.att1(att_vec)) * m.pt_mlp(nf)
top = pair.topk(min(2, pair.shape[2]), dim=2).values
if top.shape[2] == 2:
    gap = (top[:, :, 0] - top[:, :, 1])
    rel = gap / top[:, :, 0].abs().clamp_min(1e-30)
    nz = gap > 0
    for th in (1e-7, 1e-6, 1e-5, 1e-4):
        print("near-ties (0 < rel gap < %g): %d of %d" % (th, int((nz & (rel < th)).sum()), rel.numel()))
    print("exact ties with non-zero max:", int(((gap == 0) & (top[:, :, 0] != 0)).sum()))
    print("smallest positive gaps:", gap[nz].flatten().sort().values[:5].tolist())
