import copy, torch
from grid_gcn_amd import model_cls, ops, train_ops, _lib
from grid_gcn_amd.ops import _ptr, _stream
DEV = "cuda:0"
torch.manual_seed(0)
gen = torch.Generator().manual_seed(1)
cin, pt, att, O, P = 0, [64, 64, 128], [64, 128, 128], 40, 64
B, Nsrc = 3, 150
ref = model_cls.SubGUpdateCls(cin, pt, att).to(DEV).eval()
from tests.test_gpu_gridconv import randomise_bn
randomise_bn(ref.cpu(), gen); ref = ref.to(DEV)
new = copy.deepcopy(ref); ref.mfma_train = False
src = (torch.rand(B, Nsrc, 4 + cin, generator=gen) * 2 - 1).to(DEV)
nebidx = torch.randint(0, Nsrc, (B, O, P), generator=gen, dtype=torch.int32).to(DEV)
cent = (torch.rand(B, O, 4, generator=gen) * 2 - 1).to(DEV)
lib = _lib.load()
with torch.no_grad():
    nb = ops.batch_take_g(src, nebidx)
    geo = nb[..., :3] - cent[:, :, None, :3]
    dist = geo.pow(2).sum(-1, keepdim=True).sqrt()
    nf = ref.pt_mlp(geo); a1 = ref.att1(torch.cat([dist, geo], -1))
    ctx = geo.max(2, keepdim=True).values.expand_as(geo)
    att_ = ref.att2(torch.cat([a1, nf, ctx], -1))
    want = (att_ * nf).max(2).values
    E = B * O * P
    x0 = torch.empty((E, 8), device=DEV); att16 = torch.empty((E, 16), device=DEV)
    st = _stream(src)
    lib.gridgcn_edge_inputs_rows(_ptr(src), _ptr(nebidx), _ptr(cent), 4, B, Nsrc, 4, O, P, 0, 0, 8, _ptr(x0), _ptr(att16), st)
    print("x0 geo", float((x0[:, :3] - geo.reshape(E, 3)).abs().max()), "att16", float((att16[:, :4] - torch.cat([dist, geo], -1).reshape(E, 4)).abs().max()))
    Zl, scl, shl = train_ops._chain_eval_raw(lib, x0, list(new.pt_mlp))
    print("nf", float((torch.relu(Zl * scl + shl) - nf.reshape(E, -1)).abs().max()))
    Za1, sc1, sh1 = train_ops._chain_eval_raw(lib, att16, list(new.att1))
    print("a1", float((torch.relu(Za1 * sc1 + sh1) - a1.reshape(E, -1)).abs().max()))
    got = new.forward_src(cent, src, nebidx, None)
    print("agg", float((got - want).abs().max()), float(want.abs().max()))
    z20 = ref.att2[0].lin(torch.cat([a1, nf, ctx], -1)).reshape(E, -1)
    ctxv = torch.empty((B * O, 3), device=DEV)
    lib.gridgcn_ctx_max(_ptr(src), _ptr(nebidx), _ptr(cent), 4, B, Nsrc, 4, O, P, _ptr(ctxv), None, st)
    print("ctx", float((ctxv - geo.max(2).values.reshape(-1, 3)).abs().max()))
