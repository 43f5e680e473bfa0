"""`with emulated_gpu():` -- run the product's HOST code (grid_gcn_amd.ops / train / gridconv / model: the Python that
sizes workspaces, packs operands and sequences the kernels of a step) on CPU tensors, with every library call landing
in the host-side emulation of the kernels (tests/simt/).  TEST INFRASTRUCTURE: nothing here is reachable from the
package; the patches below live for the duration of the `with` block inside a test process.

What is patched, and why it is enough:
  * grid_gcn_amd._lib._lib      -> the emulated library with the prototypes of include/gridgcn.h (same entries, same
                                   argument checks: it is gridgcn_capi.hip itself, compiled for the host);
  * torch.cuda.current_stream   -> an object whose .cuda_stream is 0 (the emulator ignores streams: launches are
                                   synchronous and in program order, which is the order a single stream gives);
  * torch.cuda.device / synchronize / stream / Stream / Event -> no-ops (one stream, every wait satisfied);
  * torch.empty                 -> poisoned (NaN): reads of memory nobody wrote become visible;
  * torch.Tensor.is_cuda        -> True: the host code asks it to choose the kernel path over the stock modules, and
                                   raises on CPU tensors by design (the product has no CPU fallback -- and still has
                                   none: the tensors are CPU tensors only because the "device" is this emulator).
Every tensor stays a CPU tensor for PyTorch itself (device == cpu, allocations, autograd)."""
import contextlib
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from grid_gcn_amd import _lib  # noqa: E402

_EMU = None


def library():
    """the emulated library, built on demand, with the C ABI's prototypes attached"""
    global _EMU
    if _EMU is None:
        import build as simt_build
        # GG_SIMT_ASAN=1: the AddressSanitizer build (the process must have been started with the runtime preloaded:
        # tests/simt/asan.sh)
        lib = ctypes.CDLL(simt_build.build(asan=os.environ.get("GG_SIMT_ASAN") == "1"))
        _EMU = _lib.declare(lib, "tests/simt emulation")
        _EMU.simt_counters.argtypes = [ctypes.c_void_p]
    return _EMU


def counters():
    """(launches, rendezvous, cross-lane reads outside the executing set, rendezvous in divergent control flow,
    buffer dwords outside their descriptor's range)"""
    out = (ctypes.c_longlong * 5)()
    library().simt_counters(out)
    return tuple(out)


class _Stream:
    """launches are synchronous and in program order: every stream is the same stream, every wait is satisfied"""
    cuda_stream = 0

    def __init__(self, device=None, **kw):
        pass

    def wait_stream(self, other):
        pass

    def wait_event(self, ev):
        pass

    def synchronize(self):
        pass


class _Event:
    def __init__(self, *a, **kw):
        pass

    def record(self, stream=None):
        pass

    def synchronize(self):
        pass


_PATCHED = ("current_stream", "device", "synchronize", "Stream", "Event", "stream")


@contextlib.contextmanager
def emulated_gpu(poison=True):
    """poison: every torch.empty() of the block comes back filled with NaN (integers: a large negative number), so a
    kernel that reads what nobody wrote shows up in its results -- on the GPU fresh blocks usually hold finite
    leftovers and such a read goes unnoticed."""
    lib = library()
    saved_lib = _lib._lib
    saved = {k: getattr(torch.cuda, k) for k in _PATCHED}
    saved_empty = torch.empty
    had = "is_cuda" in torch.Tensor.__dict__

    def empty(*a, **k):
        t = saved_empty(*a, **k)
        if t.numel():
            if t.is_floating_point():
                t.fill_(float("nan"))
            elif t.dtype in (torch.int32, torch.int64):
                t.fill_(-0x5A5A5A5A)
            elif t.dtype == torch.uint8:
                t.fill_(0xA5)
        return t

    try:
        _lib._lib = lib
        torch.cuda.current_stream = lambda device=None: _Stream()
        torch.cuda.device = lambda device: contextlib.nullcontext()
        torch.cuda.synchronize = lambda *a, **k: None
        torch.cuda.Stream = _Stream
        torch.cuda.Event = _Event
        torch.cuda.stream = lambda s: contextlib.nullcontext()
        torch.Tensor.is_cuda = property(lambda self: True)
        if poison:
            torch.empty = empty
        yield lib
    finally:
        _lib._lib = saved_lib
        for k, v in saved.items():
            setattr(torch.cuda, k, v)
        torch.empty = saved_empty
        if not had:
            del torch.Tensor.is_cuda
