"""Regression guard for DESIGN 3.5 (y): the weight-staging helper must keep its eight loads in flight in the ISA.

A load whose only use sits behind a bounds check is sunk into that branch by the compiler (load, wait, store --
eight times); the first "batched" version of the helper compiled to exactly that and nobody noticed until the
assembly was read.  This test compiles a ten-line kernel around gg_stage_copy4 for gfx950 (no GPU needed) and
checks that eight vector loads are issued before the first wait on any of them."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"

SRC = r"""
#include <hip/hip_runtime.h>
#include "gridgcn_mma.h"
extern "C" __global__ __launch_bounds__(512) void k_stage(const float4 *w, int n, int stride, float *out)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    gg_stage_copy4((float4 *)lds, w, n, stride, threadIdx.x, blockDim.x);
    __syncthreads();
    out[threadIdx.x] = lds[threadIdx.x * 3];
}
"""


@pytest.mark.skipif(not os.path.exists(HIPCC) or shutil.which("c++filt") is None, reason="needs hipcc")
def test_stage_copy_keeps_eight_loads_in_flight(tmp_path):
    src = tmp_path / "k.hip"
    src.write_text(SRC)
    asm = tmp_path / "k.s"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        "-I" + os.path.join(ROOT, "grid_gcn_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
                        str(src), "-o", str(asm)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    body = [l for l in asm.read_text().splitlines() if not l.strip().startswith(";")]
    start = next(i for i, l in enumerate(body) if l.startswith("k_stage:"))
    ops = [l.strip() for l in body[start:] if re.search(r"global_load_dwordx4|s_waitcnt vmcnt|ds_write_b128", l)]
    first_wait = next(i for i, l in enumerate(ops) if l.startswith("s_waitcnt vmcnt"))
    loads_before = sum(1 for l in ops[:first_wait] if "global_load_dwordx4" in l)
    assert loads_before == 8, ops[:12]
    # and the waits count down (vmcnt(7) .. vmcnt(0)): every store waits for its own load only
    waits = [int(re.search(r"vmcnt\((\d+)\)", l).group(1)) for l in ops[first_wait:first_wait + 16]
             if l.startswith("s_waitcnt vmcnt")]
    assert waits[:8] == [7, 6, 5, 4, 3, 2, 1, 0], waits
