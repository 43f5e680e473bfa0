"""bench.py's main() with the emulator in place of cuda:0 (no GPU needed): the whole driver-facing program -- argument
handling, the training step, the timing brackets, the micro-benchmarks, the JSON line -- EXECUTED at a batch / cloud size
the emulator can afford.  Nothing it prints is a measurement (a "millisecond" here is the emulation's wall clock); what it
shows is that the program runs to its last line and that the line has every field.  TEST INFRASTRUCTURE.

    python tools/simt_bench.py cfg3 --batch 1 --points 1024 --no-micro           # bench_configs.py's path, ~4 minutes
    python tools/simt_bench.py cfg4 --batch 1 --points 1024 --micro-iters 1      # the headline path with its micro-benchmarks

Two ranks over gloo (the N > 1 path: FlatGradAllReduce, ranks_agree, the max over ranks, param_sync_spread):
    for r in 0 1; do MASTER_ADDR=127.0.0.1 MASTER_PORT=29731 WORLD_SIZE=2 RANK=$r LOCAL_RANK=$r GG_DIST_BACKEND=gloo \
        python tools/simt_bench.py cfg3 --gpus 2 --batch 1 --points 1024 --no-micro & done; wait
Eager unless --graph (a stand-in for torch.cuda.CUDAGraph whose capture runs its body once and whose replay does
nothing: graph.py's GraphedTrainStep and bench.py's graph branch then execute -- four training steps); never the CPU
baseline."""
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from simt import emu  # noqa: E402


class _Event(emu._Event):
    t = 0.0

    def record(self, stream=None):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return max((other.t - self.t) * 1e3, 1e-6)


class _TorchProxy(types.ModuleType):
    """`torch` as bench.py sees it: every device is the CPU"""

    def __init__(self):
        super().__init__("torch_proxy")

    def __getattr__(self, k):
        return getattr(torch, k)

    @staticmethod
    def device(*a, **k):
        return torch.device("cpu")


class _FakeGraph:
    """--graph: stands in for torch.cuda.CUDAGraph so that graph.py's GraphedTrainStep and bench.py's graph branch EXECUTE:
    the "capture" runs its body once, eagerly (as a capture does on the host side); a replay does nothing -- the step a
    replay would repeat has been run by the capture, which is all that can be checked without a GPU"""
    replays = 0

    def replay(self):
        _FakeGraph.replays += 1


def run(argv, graph=False):
    """bench.main() under emulated_gpu(); returns the JSON line it printed"""
    import contextlib
    import io
    import json
    import bench
    import bench_configs
    import test_simt_product as P
    px = _TorchProxy()
    saved = (bench.torch, bench_configs.torch, torch.cuda.is_available, torch.cuda.set_device, torch.cuda.device_count,
             sys.argv)
    bench.torch = bench_configs.torch = px
    torch.cuda.is_available = lambda: True
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.device_count = lambda: 1
    sys.argv = ["bench.py"] + list(argv) + ["--steps", "1", "--warmup", "0", "--no-cpu-baseline"] + ([] if graph else ["--eager"])
    saved_graph = (torch.cuda.CUDAGraph, torch.cuda.graph)
    if graph:
        torch.cuda.CUDAGraph = _FakeGraph
        torch.cuda.graph = lambda g, **kw: contextlib.nullcontext()
    buf = io.StringIO()
    try:
        with emu.emulated_gpu(poison=False):
            torch.cuda.Event = _Event
            with contextlib.redirect_stdout(buf):
                P._to_cpu(bench.main)()
    finally:
        (bench.torch, bench_configs.torch, torch.cuda.is_available, torch.cuda.set_device, torch.cuda.device_count,
         sys.argv) = saved
        torch.cuda.CUDAGraph, torch.cuda.graph = saved_graph
    out = buf.getvalue().strip().splitlines()
    return json.loads(out[-1]) if out else None       # (a rank other than 0 prints nothing)


if __name__ == "__main__":
    t0 = time.time()
    args = sys.argv[1:]
    use_graph = "--graph" in args
    args = [x for x in args if x != "--graph"]
    if args and not args[0].startswith("-"):
        args = ["--config", args[0]] + args[1:]
    line = run(args, graph=use_graph)
    import json
    if line is not None:
        print(json.dumps(line))
    print("bench.main() on the emulator: ran to its last line in %.0f s; keys: %s" % (
        time.time() - t0, sorted(line) if line else "(no line: not rank 0)"), file=sys.stderr)
