"""A whole training step as ONE hipGraph (torch.cuda.CUDAGraph on ROCm).

A step of the segmentation network is ~350 launches (hand-written kernels through ctypes + a few
framework ops); enqueueing them costs the host 7-12 ms depending on the box -- as long as a step of
the GPU.  Every launch of this library goes to the caller's stream, takes its scratch from the
caching allocator and never synchronises, so the step is capturable as it is:
replaying the graph removes the host from the timed path altogether.

What must not be frozen into a graph are the per-step random draws.  The reference reseeds its
samplers at every operator call (gridify.cu:377-379); here every sampling / dropout seed inside the
kernels is `seed + *seed_dev` (include/gridgcn.h: gridgcn_grid_params.seed_dev, drop_seed_dev), and
the captured graph starts with `seed_dev += golden ratio`: each replay redraws the voxel sampling,
the neighbour reservoirs and the dropout mask.

Data parallelism: with world_size > 1 the step is two graphs -- forward/backward, then the
optimizer -- around the ONE flat RCCL all-reduce of dp.FlatGradAllReduce, which stays eager (a
collective captured into a graph is the only part of this that cannot be exercised here).
"""
import torch

_GOLDEN = 0x9E3779B97F4A7C15 - (1 << 64)   # as a signed int64 increment


class GraphedTrainStep:
    """step() == zero_grad(set_to_none) ; loss = loss_fn(net(*inputs), target) ; backward ;
    all-reduce (world > 1) ; opt.step() -- replayed from captured graphs.

    net       a GGCNSeg / GGCNCls / GGCNSynth in train() mode on the GPU
    opt       torch.optim.Adam(..., fused=True, capturable=True) (any capturable optimizer)
    inputs    tuple of STATIC input tensors (their storage is what the graph reads: refill them
              in place to feed a new batch); target likewise
    sync      dp.FlatGradAllReduce or None
    """

    def __init__(self, net, opt, loss_fn, inputs, target, sync=None, warmup=3):
        self.net, self.opt, self.loss_fn = net, opt, loss_fn
        self.inputs, self.target, self.sync = tuple(inputs), target, sync
        dev = self.inputs[0].device
        self.world = sync.world if sync is not None else 1
        net.seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self.params = [p for p in net.parameters() if p.requires_grad]

        def fwd_bwd():
            net.seed_dev.add_(_GOLDEN)
            opt.zero_grad(set_to_none=True)
            loss = loss_fn(net(*self.inputs), self.target)
            loss.backward()
            return loss.detach()

        # warm-up on a side stream (allocator pools, lazy initialisations, autotuned paths)
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(warmup):
                fwd_bwd()
                if self.world > 1:
                    sync()
                opt.step()
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)

        self.g1 = torch.cuda.CUDAGraph()
        self.g2 = None
        if self.world == 1:
            with torch.cuda.graph(self.g1):
                self.loss = fwd_bwd()
                opt.step()
        else:
            pool = torch.cuda.graph_pool_handle()
            with torch.cuda.graph(self.g1, pool=pool):
                self.loss = fwd_bwd()
            # the gradients the captured backward writes (graph-owned storage)
            self.static_grads = [p.grad for p in self.params]
            assert all(g is not None for g in self.static_grads)
            self._allreduce()                      # p.grad become views of the flat bucket
            self.g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g2, pool=pool):
                opt.step()
        torch.cuda.synchronize(dev)

    def _allreduce(self):
        sync = self.sync
        with torch.no_grad():
            torch.cat([g.reshape(-1) for g in self.static_grads], out=sync.flat)
            torch.distributed.all_reduce(sync.flat, op=torch.distributed.ReduceOp.SUM)
            if sync.average:
                sync.flat.div_(sync.world)
            off = 0
            for p in self.params:
                n = p.numel()
                p.grad = sync.flat[off:off + n].view_as(p)
                off += n

    def __call__(self):
        self.g1.replay()
        if self.g2 is not None:
            self._allreduce()
            self.g2.replay()
        return self.loss
