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

Data parallelism: with world_size > 1 the ONE graph also holds the gather of the gradients into
dp.FlatGradAllReduce's flat bucket, the RCCL all-reduce (a collective is capturable: it is a kernel
on the capture stream) and the averaging, in front of the optimizer.  An earlier arrangement -- a
forward/backward graph, the all-reduce eager, an optimizer graph -- was dropped: on this ROCm
version eager kernels that consume the output of a just-launched graph (and graphs that follow
eager kernels) are not reliably ordered on the stream; its single-rank emulation trained along a
different trajectory from run to run.  Back-to-back replays of one graph are.
tests/test_gpu_gridconv.py captures the collective on one rank over a real RCCL group (more ranks
need more GPUs) and checks the trajectory against the graph without it.  If the capture fails,
bench.py falls back to the eager step and says so (`step_mode`).
"""
import os

import torch

from .train import common as tcommon

_GOLDEN = 0x9E3779B97F4A7C15 - (1 << 64)   # as a signed int64 increment


class GraphedTrainStep:
    """step() == zero_grad(set_to_none) ; loss = loss_fn(net(*inputs), target) ; backward ;
    all-reduce (world > 1) ; opt.step() -- replayed from one captured graph.

    net       a GGCNSeg / GGCNCls / GGCNSynth in train() mode on the GPU
    opt       grid_gcn_amd.optim.Adam, or torch.optim.Adam(..., fused=True, capturable=True) (any capturable
              optimizer)
    inputs    tuple of STATIC input tensors (their storage is what the graph reads: refill them
              in place to feed a new batch); target likewise
    sync      dp.FlatGradAllReduce or None
    """

    def __init__(self, net, opt, loss_fn, inputs, target, sync=None, warmup=3, split=None):
        self.net, self.opt, self.loss_fn = net, opt, loss_fn
        self.inputs, self.target, self.sync = tuple(inputs), target, sync
        dev = self.inputs[0].device
        self.world = sync.world if sync is not None else 1
        # gradient gather + all-reduce inside the graph (default: whenever there is more than one
        # rank; split=True forces it on a single rank, for tests)
        self.split = (self.world > 1) if split is None else bool(split)
        assert not self.split or sync is not None
        net.seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self.params = [p for p in net.parameters() if p.requires_grad]
        self._state = list(net.parameters()) + [b for b in net.buffers() if b.is_floating_point()]

        # (kept alive with the graph: the captured loss-gradient kernel reads it at every replay)
        self._one = one = torch.ones((), dtype=torch.float32, device=dev)

        def fwd_bwd():
            net.seed_dev.add_(_GOLDEN)
            opt.zero_grad(set_to_none=True)
            loss = loss_fn(net(*self.inputs), self.target)
            loss.backward(one)       # (the default gradient would be one more fill launch per step)
            return loss.detach()

        # warm-up on a side stream (allocator pools, lazy initialisations, autotuned paths)
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(warmup):
                fwd_bwd()
                if self.split:
                    self._sync_eager()
                opt.step()
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)

        self.g1 = torch.cuda.CUDAGraph()
        # scratch carved out of a shared zero chunk must not cross a capture boundary
        tcommon.reset_zero_arena()
        # capture_error_mode "thread_local": the RCCL watchdog thread polls its events with
        # hipEventQuery while this thread captures; under the default "global" mode such a call from
        # ANY thread invalidates the capture ("operation not permitted when stream is capturing": seen
        # in about one of three single-rank runs, whenever a poll fell inside the ~1 s capture)
        with torch.cuda.graph(self.g1, capture_error_mode="thread_local"):
            if os.environ.get("GG_TEST_CAPTURE_FAIL") == "1":
                torch.cuda.synchronize(dev)        # test hook: an illegal call invalidates the capture
            self.loss = fwd_bwd()
            if self.split:
                # the flat gather, the RCCL all-reduce and the averaging are graph nodes too
                self.static_grads = [p.grad for p in self.params]
                assert all(g is not None for g in self.static_grads)
                self._allreduce()                  # p.grad become views of the flat bucket
            opt.step()
        tcommon.reset_zero_arena()
        torch.cuda.synchronize(dev)

    def _sync_eager(self):
        """warm-up steps: the plain FlatGradAllReduce call (a no-op on one rank, where the forced
        split still has to leave .grad as views of the flat bucket)"""
        if self.sync.world > 1:
            self.sync()
        else:
            self.static_grads = [p.grad for p in self.params]
            self._allreduce()

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
        # a replay rewrites parameters and BatchNorm buffers without any Python running: caches keyed on
        # Tensor._version (folded evaluation constants, gridconv.SubGUpdate.packed_layers) must see it
        with torch.no_grad():
            torch.autograd.graph.increment_version(self._state)
        tcommon.params_changed()
        return self.loss
