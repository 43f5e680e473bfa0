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
        # GG_SIMT_RACE=1: the data-race detector (tests/simt/simt_race.cpp); its report goes to the file named by
        # GG_SIMT_RACE_REPORT when the process ends (tests/simt/race.sh)
        race = os.environ.get("GG_SIMT_RACE") == "1"
        # GG_SIMT_UBSAN=1: the UndefinedBehaviorSanitizer build (tests/simt/ubsan.sh)
        lib = ctypes.CDLL(simt_build.build(asan=os.environ.get("GG_SIMT_ASAN") == "1", race=race,
                                           ubsan=os.environ.get("GG_SIMT_UBSAN") == "1"))
        _EMU = _lib.declare(lib, "tests/simt emulation")
        _EMU.simt_counters.argtypes = [ctypes.c_void_p]
        _EMU.simt_set_order_filter.argtypes = [ctypes.c_char_p]
        # GG_SIMT_ORDER=<bit mask>: every launch of the process in another schedule (tests/simt/simt_hip.h: simt_order)
        if os.environ.get("GG_SIMT_ORDER"):
            _EMU.simt_set_order(int(os.environ["GG_SIMT_ORDER"]))
        if race:
            import atexit
            atexit.register(_write_race_report)
        # GG_SIMT_COVERAGE=<file>: the kernels this process launched are appended there when it ends
        if os.environ.get("GG_SIMT_COVERAGE"):
            import atexit
            atexit.register(_write_coverage)
    return _EMU


def kernel_coverage():
    """{kernel expression: launches} of this process"""
    lib = library()
    lib.simt_kernel_coverage.restype = ctypes.c_int
    lib.simt_kernel_coverage.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    buf = ctypes.create_string_buffer(1 << 20)
    lib.simt_kernel_coverage(buf, len(buf))
    return {ln.split("\t")[0]: int(ln.split("\t")[1]) for ln in buf.value.decode().splitlines() if ln}


def _write_coverage():
    path = os.environ["GG_SIMT_COVERAGE"]
    with open(path, "a") as f:
        for k, v in sorted(kernel_coverage().items()):
            f.write("%s\t%d\n" % (k, v))


def race_report(lib=None):
    """[(kind, kernel, file:line of the earlier access, file:line of the later one, count, example)] of the race build
    (or of another library with the detector linked in)"""
    import subprocess
    lib = lib or library()
    lib.simt_race_report.restype = ctypes.c_int
    lib.simt_race_report.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    buf = ctypes.create_string_buffer(1 << 22)
    lib.simt_race_report(buf, len(buf))
    rows = [ln.split("\t") for ln in buf.value.decode().splitlines() if ln]
    if not rows:
        return []
    so = next((r[2] for r in rows if r[2] != "?"), rows[0][2])
    addrs = [a for r in rows for a in (r[3], r[4])]
    # (the address of the call's return: one byte back is inside the access's own line)
    sym = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-symbolizer", "--obj=" + so, "--functions=none", "--no-inlines"] +
                         [hex(int(a, 16) - 1) for a in addrs], capture_output=True, text=True).stdout
    locs = [ln.strip() for ln in sym.splitlines() if ln.strip()]

    def hip_line(l):
        # generated unit -> the product's source: one header line was added in front (tests/simt/build.py)
        f, _, rest = os.path.basename(l).partition(":")
        if f.endswith(".simt.cpp") and rest.split(":")[0].isdigit():
            return "%s:%d" % (f.replace(".simt.cpp", ".hip"), int(rest.split(":")[0]) - 1)
        if rest.split(":")[0].isdigit():
            return "%s:%s" % (f, rest.split(":")[0])
        return os.path.basename(l)

    locs = [hip_line(l) for l in locs]
    agg = {}        # template instantiations and inlined copies of one source line: one row
    for i, r in enumerate(rows):
        a, b = (locs[2 * i], locs[2 * i + 1]) if len(locs) >= 2 * i + 2 else (r[3], r[4])
        same = r[6].startswith("same-value stores: ")
        k = (r[0], r[1], a, b)
        if k in agg:
            n, ex, sm = agg[k]
            agg[k] = (n + int(r[5]), ex if sm or not same else r[6], sm and same)
        else:
            agg[k] = (int(r[5]), r[6], same)
    return [(k[0], k[1], k[2], k[3], v[0], v[1] if v[2] or not v[1].startswith("same-value") else
             v[1][len("same-value stores: "):]) for k, v in sorted(agg.items())]


def race_counters():
    """(instrumented accesses checked, conflicts seen, distinct reports)"""
    out = (ctypes.c_longlong * 3)()
    library().simt_race_counters(out)
    return tuple(out)


def _write_race_report():
    path = os.environ.get("GG_SIMT_RACE_REPORT", "/tmp/simt_race_report.txt")
    rows = race_report()
    with open(path, "w") as f:
        f.write("# accesses checked %d, conflicts %d, distinct %d\n" % race_counters())
        for r in rows:
            f.write("%s\t%s\t%s\t%s\t%d\t%s\n" % r)


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


_PATCHED = ("current_stream", "device", "synchronize", "Stream", "Event", "stream", "is_current_stream_capturing")


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
        torch.cuda.is_current_stream_capturing = lambda: False
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
