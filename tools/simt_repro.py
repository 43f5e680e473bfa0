"""Run-to-run reproducibility of the training kernels, asked of the emulator (no GPU needed):

    python tools/simt_repro.py  ->  profiles/r6_repro.txt

The GPU runs the workgroups of a launch in an order nobody controls; the emulator (tests/simt) can CHOOSE it.  An MLP
chain (conv + BatchNorm + ReLU x 3, forward + backward) and an up layer of the segmentation net (the Z2-free attention
pair, pair product / max, sparse backward) are run on the same inputs with the workgroups / waves / lanes ascending,
descending and permuted; every output, input gradient and parameter gradient is compared BIT FOR BIT with the ascending
run.  What differs does so through floating-point atomics (BatchNorm sums: fp64 atomicAdd of fp32 partials) -- the
table says which tensors that reaches and by how much."""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from simt import emu  # noqa: E402
import test_simt_product as P  # noqa: E402

ORDERS = [(0, "ascending"), (7, "workgroups, waves, lanes descending"), (10, "workgroups permuted, waves descending"),
          (9, "workgroups permuted + descending")]


def mlp_case():
    import test_gpu_train_ops as T
    from grid_gcn_amd.train import mlp as tmlp
    torch.manual_seed(3)
    net = T.mlp(72, [64, 128, 128]).train()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.3)
    state = copy.deepcopy(net.state_dict())
    x0 = torch.randn(5000, 72) * 1.5
    cot = torch.randn(5000, 128)

    def run():
        net.load_state_dict(state)
        net.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        y = tmlp.mlp_bn_relu_train(x, list(net))
        (y * cot).sum().backward()
        out = {"y": y.detach().clone(), "dx": x.grad.clone()}
        for n, p in net.named_parameters():
            out["d " + n] = p.grad.clone()
        for n, b in net.named_buffers():
            if b.dtype.is_floating_point:
                out["buf " + n] = b.clone()
        return out
    return run


def up_case():
    from grid_gcn_amd import ops  # noqa: F401
    layer, src, upl, nebidx = P._up_layer_case(5, B=2, Nsrc=200, O=1500)
    state = copy.deepcopy(layer.state_dict())
    cot = torch.randn(src.shape[0], nebidx.shape[1], 128)

    def run():
        layer.load_state_dict(state)
        layer.zero_grad(set_to_none=True)
        s = src.clone().requires_grad_(True)
        y = layer.forward_src(upl, s, nebidx, None, center_ori_feats=upl)
        (y * cot).sum().backward()
        out = {"y": y.detach().clone(), "d src feats": s.grad[..., 4:].clone()}
        for n, p in layer.named_parameters():
            out["d " + n] = p.grad.clone()
        return out
    return run


def main():
    lines = ["# bit-for-bit comparison with the ascending schedule (tools/simt_repro.py; emulator, no GPU)",
             "# per tensor: identical, or max |difference| relative to max |value|"]
    lib = emu.library()
    for name, make in (("MLP chain 72 -> 64 -> 128 -> 128, 5000 rows", mlp_case), ("up layer, 2 x 1500 centres x 5", up_case)):
        with emu.emulated_gpu():
            run = make()
            base = None
            for order, what in ORDERS:
                lib.simt_set_order(order)
                out = run()
                lib.simt_set_order(0)
                if base is None:
                    base = out
                    continue
                diff = []
                for k in base:
                    if not torch.equal(base[k], out[k]):
                        s = float(base[k].abs().max()) or 1.0
                        diff.append("%s %.1e" % (k, float((base[k] - out[k]).abs().max()) / s))
                lines.append("%-46s %-40s %d of %d tensors identical%s" % (
                    name, what, len(base) - len(diff), len(base), "" if not diff else "; differ: " + ", ".join(diff)))
    txt = "\n".join(lines) + "\n"
    open(os.path.join(ROOT, "profiles", "r6_repro.txt"), "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
