"""The forward of the up layers' attention pair without its [E, 128] pre-activation (csrc/gridgcn_attfwd.hip), pinned
in the gfx950 ISA (CPU tier: hipcc cross-compiles).  What the kernel's speed rests on are compiler decisions a
toolchain change could silently undo; each cost 0.1-0.25 ms of the 0.45 ms kernel when it went wrong
on the way:
  * no scratch: a spilled register is reloaded with a VMEM instruction whose s_waitcnt vmcnt(0) drains every gather in
    flight;
  * the MFMA accumulators live in ordinary VGPRs (amdgpu_waves_per_eu): with the default bound they sit in AGPRs and
    every value the fold touches costs a v_accvgpr_read;
  * the gathers are RAW buffer loads (the structured form -- row index as vindex -- measured +0.24 ms);
  * 64 MFMAs per tile (four column tiles x sixteen steps), the point branch on packed fp32 instructions;
  * the moments loop carries no predicate and no AGPR traffic."""
import os
import re
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import isa  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(isa.HIPCC) or shutil.which("c++filt") is None,
                                reason="needs hipcc")


def _ops(body):
    return [l.strip() for l in body if l.strip() and not l.strip().endswith(":") and not l.strip().startswith(".")]


def _loops(body):
    """instruction lists of the backward-branch loops of a kernel body, longest first"""
    lab, out = {}, []
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            lab[m.group(1)] = i
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch\S*\s+(\.LBB\d+_\d+)", l)
        if m and lab.get(m.group(1), 1 << 30) < i:
            out.append(_ops(body[lab[m.group(1)]:i]))
    return sorted(out, key=len, reverse=True)


def test_att_pairmax_tile_loop():
    ks = isa.kernels("gridgcn_attfwd.hip")
    body = ks["gg_k_att_pairmax<true>"]
    # (the evaluation form <false>: the same loop without arg max and saved pre-activations)
    ev = _loops(ks["gg_k_att_pairmax<false>"])[0]
    assert sum(1 for o in ev if o.startswith("v_mfma_f32_32x32x2")) == 64 and len(ev) <= 900, len(ev)
    assert not any(o.startswith(("scratch_", "v_accvgpr", "buffer_store_byte")) for o in ev)
    ops = _ops(body)
    assert not any(o.startswith("scratch_") for o in ops)
    assert not any(o.startswith("v_accvgpr") for o in ops)
    loop = _loops(body)[0]
    assert sum(1 for o in loop if o.startswith("v_mfma_f32_32x32x2")) == 64
    gathers = [o for o in loop if o.startswith("buffer_load_dword ")]
    assert len(gathers) >= 64 and not any("idxen" in o for o in gathers)
    # the next tile's rows (four 16-byte loads) and edge record are requested inside the loop: the prefetch
    assert sum(1 for o in loop if o.startswith("buffer_load_dwordx4")) >= 4
    # outputs leave through range-checked buffer stores, no predicate branch around them
    assert sum(1 for o in loop if o.startswith("buffer_store_dword")) == 36
    assert sum(1 for o in loop if o.startswith("buffer_store_byte")) == 12
    assert sum(1 for o in loop if o.startswith("v_pk_fma_f32")) >= 60
    # and the whole tile stays an instruction-count budget (1124 when written; the kernel is issue bound)
    assert len(loop) <= 1250, len(loop)


def test_att_moments_main_loop():
    body = isa.kernels("gridgcn_attfwd.hip")["gg_k_att_moments"]
    assert not any(o.startswith(("scratch_", "v_accvgpr")) for o in _ops(body))
    main = [lp for lp in _loops(body) if sum(1 for o in lp if o.startswith("v_mfma")) == 16]
    assert main, "the 16-step batch loop"
    lp = main[0]
    # (two compares are the loop control: the batch count is a per-wave value)
    assert not any(o.startswith("v_cndmask") for o in lp) and sum(1 for o in lp if o.startswith("v_cmp")) <= 2
    assert len(lp) <= 100, len(lp)
