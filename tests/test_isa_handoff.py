"""The inter-workgroup hand-offs of the training kernels, pinned in the gfx950 ISA (CPU tier: hipcc cross-compiles).

DESIGN 3.5 (r), (t) / ADVICE r4: the folded BatchNorm finalisation (gg_k_linear_fwd_direct), the dW slice reduce and
the loss / column-sum finish hand data from many workgroups to the last arriver WITHOUT an agent-scope release fence
(`buffer_wbl2` = a write-back of the XCD's whole L2 per workgroup: 45 us instead of 22 for the loss kernel).  That is
the "{agent-scope atomics / sc1 accesses on both sides, drained before a relaxed ticket}" form of MI355X_MICROARCH --
valid on gfx950, outside what the HIP memory model promises, so a compiler change could silently break it.  This test
reads the assembly and checks the instructions the argument rests on:
  producer   the data leave through L2 atomics or sc1 (write-through) stores, `s_waitcnt vmcnt(0)` + `s_barrier`
             stand between them and the ticket, the ticket is a RETURNING agent-scope atomic add;
  consumer   the last arriver reads the slots with sc1 loads (or behind an acquire: `buffer_inv sc1`);
and that gg_k_gemm_tn keeps the release / acquire pair it was given in round 4 (the one hand-off whose data are
plain stores).  The GPU-side stress test of the same hand-offs is tests/test_gpu_handoff.py."""
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

TICKET = re.compile(r"\b(global|flat)_atomic_add(_u32)?\b.*\bsc0\b")      # returning 32-bit add
DATA_ATOMIC = re.compile(r"\b(global|flat)_atomic_add_(f64|x2)\b")


def _ops(body):
    return [l.strip() for l in body if l.strip() and not l.strip().endswith(":") and not l.strip().startswith(".")]


def _tickets(ops):
    return [i for i, l in enumerate(ops) if TICKET.search(l)]


def _drained_before(ops, t, data, barrier=True):
    """between the last data-leaving instruction in front of ticket t and the ticket: `s_waitcnt vmcnt(0)` (the
    issuing lanes' atomics / write-through stores have reached L2), then -- when other waves of the workgroup
    produced data too -- the `s_barrier` that makes the ticket thread wait for THEIR drains"""
    last = max(i for i in data if i < t)
    w = ops[last + 1:t]
    drain = [i for i, l in enumerate(w) if "vmcnt(0)" in l]
    if not drain:
        return False
    return (not barrier) or any(l.startswith("s_barrier") for l in w[drain[0]:])


def test_folded_batchnorm_finalisation_handoff():
    ks = isa.kernels("gridgcn_direct.hip")
    names = [n for n in ks if n.startswith("gg_k_linear_fwd_direct<")]
    assert len(names) >= 8
    for n in names:
        ops = _ops(ks[n])
        tk = _tickets(ops)
        assert len(tk) == 1, (n, len(tk))
        t = tk[0]
        # the statistics are L2 atomics (fp64 adds), all in front of the ticket
        data = [i for i, l in enumerate(ops) if DATA_ATOMIC.search(l)]
        assert data and max(data) < t, n
        assert _drained_before(ops, t, data), n
        # the last arriver reads them back through L2 (agent-scope atomic loads = sc1), behind a barrier
        tail = ops[t:]
        assert any(l.startswith("s_barrier") for l in tail[:24]), n
        assert sum(1 for l in tail if re.search(r"_load_dwordx2 .*\bsc1\b", l)) >= 2, n
        # and nobody pays for an L2 write-back
        assert not any("buffer_wbl2" in l for l in ops), n


def test_dw_reduce_handoff():
    ops = _ops(isa.kernels("gridgcn_direct.hip")["gg_k_dw_reduce_direct"])
    tk = _tickets(ops)
    assert len(tk) == 1
    t = tk[0]
    stores = [i for i, l in enumerate(ops[:t]) if re.search(r"_store_dword .*\bsc1\b", l)]
    assert stores, "slice sums must leave as sc1 (write-through) stores"
    assert _drained_before(ops, t, stores)
    assert any("buffer_inv sc1" in l for l in ops[t:t + 40]), "the last arriver's acquire"
    assert not any("buffer_wbl2" in l for l in ops)


def test_loss_and_colsum_finish_handoff():
    ks = isa.kernels("gridgcn_head.hip")
    seen = 0
    for n, body in ks.items():
        ops = _ops(body)
        tk = _tickets(ops)
        if not tk:
            continue
        seen += 1
        data = [i for i, l in enumerate(ops) if DATA_ATOMIC.search(l)]
        assert data and min(data) < tk[0], n
        # (one thread adds the workgroup's sums and draws the ticket: program order + its own drain)
        assert _drained_before(ops, tk[0], data, barrier=False), n
        assert any(re.search(r"_load_dword(x2)? .*\bsc1\b", l) for l in ops[tk[-1]:]), n
        assert not any("buffer_wbl2" in l for l in ops), n
    assert seen >= 2          # gg_k_ce_fwd<..>, gg_k_colsum<..>


def test_gemm_tn_keeps_release_acquire():
    ops = _ops(isa.kernels("gridgcn_gemm.hip")["gg_k_gemm_tn"])
    tk = _tickets(ops)
    assert len(tk) == 1
    t = tk[0]
    assert any("buffer_wbl2" in l for l in ops[max(0, t - 40):t]), "release in front of the ticket"
    assert any("buffer_inv sc1" in l for l in ops[t:t + 40]), "acquire of the last arriver"
