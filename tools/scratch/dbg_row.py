import sys, copy, torch
sys.path.insert(0, '.')
from grid_gcn_amd import train_ops
from grid_gcn_amd.gridconv import mlp
DEV = torch.device("cuda:0")
for E, cin, dims in ((4097, 10, [16, 64]), (4097, 16, [16, 64]), (4100, 10, [16, 64]), (4129, 10, [16, 64])):
    torch.manual_seed(E + cin)
    ref = mlp(cin, dims).to(DEV).train()
    for m in ref.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.3)
    new = copy.deepcopy(ref)
    x1 = (torch.randn(E, cin, device=DEV) * 1.5).requires_grad_(True)
    x2 = x1.detach().clone().requires_grad_(True)
    y1 = ref(x1)
    y2 = train_ops.mlp_bn_relu_train(x2, list(new))
    g = torch.randn_like(y1)
    y1.backward(g); y2.backward(g)
    s = float(x1.grad.abs().max())
    bad = ((x2.grad - x1.grad).abs().amax(dim=1) > 2e-4 * s).nonzero().flatten().tolist()
    print(E, cin, "bad rows", bad, "ydiff", float((y1 - y2).abs().max()))
    for r in bad[:3]:
        # pre-activations of the first layer for this row, from the reference
        z = ref[0].lin(x1[r:r+1])
        bn = ref[0].bn
        xs = ref[0].lin(x1)
        mu, var = xs.mean(0), xs.var(0, unbiased=False)
        pre = (z - mu) / torch.sqrt(var + bn.eps) * bn.weight + bn.bias
        print("   row", r, "min |pre-activation| layer0:", float(pre.abs().min()))
