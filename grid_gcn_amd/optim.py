"""Adam on one kernel launch (csrc/gridgcn_optim.hip).

The reference trains with mx.optimizer.Adam(learning_rate, wd, beta1, beta2)
(segmentation/train_test/base_solver.py:105-114).  A network of this family has ~130 small parameter
tensors; the framework's multi-tensor Adam spends four ~20 us launches on them, this one a single launch
whose pointer table travels in the kernel arguments.

`Adam` follows torch.optim.Adam (L2 weight decay added to the gradient, no amsgrad / maximize) to within
rounding; mxnet=True selects mx.optimizer.Adam's form of the same update (eps outside the bias
correction: w -= lr sqrt(1-b2^t)/(1-b1^t) m / (sqrt(v) + eps)).  The step count lives on the device, so
replays of a captured hipGraph advance it; lr may be a float (frozen into a captured graph) or a float32
GPU scalar tensor (read by the kernel at every launch: schedulers under a graph).
One step count per parameter GROUP: a parameter without a gradient in some step keeps its moments but
shares the group's count (torch counts per parameter; identical whenever every parameter gets a gradient
in every step).
"""
import ctypes

import torch

from . import _lib

CHUNK = 1024      # GG_ADAM_CHUNK


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, mxnet=False):
        if not (isinstance(lr, torch.Tensor) or lr >= 0.0):
            raise ValueError("invalid learning rate %r" % (lr,))
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("invalid betas %r" % (betas,))
        if eps < 0.0 or weight_decay < 0.0:
            raise ValueError("invalid eps / weight_decay")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay,
                                      mxnet=bool(mxnet)))
        self._flat = {}       # id(group) -> dict(m, v, state, slot: {id(p): first chunk}, tables: {...})

    def _group_state(self, group):
        ps = group["params"]
        fs = self._flat.get(id(group))
        if fs is not None and fs["ids"] == [id(p) for p in ps]:
            return fs
        if not ps:
            return None
        dev = ps[0].device
        for p in ps:
            if not (p.is_cuda and p.dtype == torch.float32 and p.device == dev and p.is_contiguous()):
                raise RuntimeError("grid_gcn_amd.optim.Adam: float32 contiguous parameters on one GPU only")
        slot, c = {}, 0
        for p in ps:
            slot[id(p)] = c
            c += (p.numel() + CHUNK - 1) // CHUNK
        m = torch.zeros(max(c, 1) * CHUNK, dtype=torch.float32, device=dev)
        v = torch.zeros_like(m)
        st = torch.zeros(2, dtype=torch.int32, device=dev)          # (step count, ticket)
        old = self._flat.get(id(group))
        fs = dict(m=m, v=v, state=st, slot=slot, ids=[id(p) for p in ps], tables={}, dev=dev)
        for p in ps:
            o, n = slot[id(p)] * CHUNK, p.numel()
            s = self.state[p]
            if "exp_avg" in s:                  # (loaded from a checkpoint, or the group changed)
                m[o:o + n].copy_(s["exp_avg"].reshape(-1))
                v[o:o + n].copy_(s["exp_avg_sq"].reshape(-1))
                if "step" in s:
                    st[0] = int(s["step"])
            s["exp_avg"] = m[o:o + n].view_as(p)
            s["exp_avg_sq"] = v[o:o + n].view_as(p)
            s["step"] = st[0:1]
        del old
        self._flat[id(group)] = fs
        return fs

    def state_dict(self):
        """torch.optim.Adam's schema.  Inside this optimizer every parameter's `step` is a view of the GROUP's one
        device-side counter; a checkpoint carries an independent float32 scalar per parameter instead (what
        torch.optim.Adam stores), so that loading it there does not advance one shared element once per parameter."""
        sd = super().state_dict()
        # (torch hands out the optimizer's OWN per-parameter dicts: the replacement goes into copies, never into
        #  self.state[p], whose `step` must stay a view of the live counter)
        sd["state"] = {k: ({**s, "step": s["step"].detach().to(torch.float32).reshape(()).clone()}
                           if "step" in s else dict(s)) for k, s in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for g in self.param_groups:
            g.setdefault("mxnet", False)     # (a torch.optim.Adam checkpoint has no such key)
        self._flat.clear()        # the loaded moments are copied into fresh flat buffers at the next step

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for group in self.param_groups:
            fs = self._group_state(group)
            if fs is None:
                continue
            ps = [p for p in group["params"] if p.grad is not None]
            for p in ps:
                g = p.grad
                if g.is_sparse or g.dtype != torch.float32 or g.device != p.device:
                    raise RuntimeError("grid_gcn_amd.optim.Adam: dense float32 gradients only")
                if not g.is_contiguous():
                    p.grad = g = g.contiguous()
            key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in ps)
            tb = fs["tables"].get(key)
            if tb is None:
                n = len(ps)
                vp, ll = ctypes.c_void_p * max(n, 1), ctypes.c_longlong * max(n, 1)
                tb = (vp(*[p.data_ptr() for p in ps]), vp(*[p.grad.data_ptr() for p in ps]),
                      ll(*[p.numel() for p in ps]), ll(*[fs["slot"][id(p)] for p in ps]), n)
                if len(fs["tables"]) > 8:
                    fs["tables"].clear()
                fs["tables"][key] = tb
            lr = group["lr"]
            lr_dev = None
            if isinstance(lr, torch.Tensor):
                if not (lr.is_cuda and lr.dtype == torch.float32 and lr.numel() == 1):
                    raise RuntimeError("grid_gcn_amd.optim.Adam: a tensor lr must be a float32 GPU scalar")
                lr_dev, lr = ctypes.c_void_p(lr.data_ptr()), 0.0
            dev = fs["dev"]
            with torch.cuda.device(dev):
                rc = lib.gridgcn_adam_step(tb[0], tb[1], tb[2], tb[3], tb[4],
                                           ctypes.c_void_p(fs["m"].data_ptr()),
                                           ctypes.c_void_p(fs["v"].data_ptr()),
                                           ctypes.c_void_p(fs["state"].data_ptr()), float(lr), lr_dev,
                                           float(group["betas"][0]), float(group["betas"][1]),
                                           float(group["eps"]), float(group["weight_decay"]),
                                           1 if group.get("mxnet", False) else 0,
                                           torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(rc, "gridgcn_adam_step")
            # (the kernel writes through raw pointers: caches keyed on Tensor._version -- the folded
            #  evaluation weights of gridconv.SubGUpdate.packed_layers -- must see the update)
            torch.autograd.graph.increment_version(ps)
        return loss
