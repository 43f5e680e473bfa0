"""The data-race detector of the CPU build (tests/simt/simt_race.cpp: clang's ThreadSanitizer instrumentation of the
kernel sources, the emulator's own notion of what orders two accesses as the runtime) on kernels with KNOWN races:
every broken form must be reported under the right kind and source line, every repaired form must be silent.
TEST INFRASTRUCTURE about test infrastructure -- the product's kernels go through the same detector in
tests/simt/race.sh (profiles/r6_race_report.txt)."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "simt"))

SRC = open(os.path.join(HERE, "simt", "race_selftest.hip")).read().splitlines()


def line_of(pattern, nth=0):
    hits = [i + 1 for i, l in enumerate(SRC) if re.search(pattern, l)]
    return "race_selftest.hip:%d" % hits[nth]


@pytest.fixture(scope="module")
def lib():
    import build as simt_build
    so = ctypes.CDLL(simt_build.build_selftest())
    so.rk_run.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3
    so.rk_run.restype = None
    return so


def run(lib, which, arg=0, uninit=False):
    """the report of one launch; UNINIT rows (a load that meets no store of its workgroup: every load that runs ahead of
    the store it races with is one, too) only on request"""
    import emu
    a, b, c = (np.zeros(4096, np.int32) for _ in range(3))
    lib.simt_race_reset()
    lib.rk_run(which, arg, a.ctypes.data, b.ctypes.data, c.ctypes.data)
    rep = emu.race_report(lib)
    return [r for r in rep if uninit or r[0] != "UNINIT"], (a, b, c)


def test_missing_barrier_between_waves(lib):
    rep, (out, _, _) = run(lib, 0, 0)
    st, ld = line_of(r"buf\[t\] = t \+ 1;"), line_of(r"out\[blockIdx.x \* 128 \+ t\] = buf")
    # (wave 0 runs first here: its loads meet wave 1's later stores, wave 1's loads meet wave 0's earlier stores)
    assert {(r[0], r[2], r[3]) for r in rep} == {("INTRA", st, ld), ("INTRA", ld, st)}, rep
    assert all("lds " in r[5] for r in rep)
    rep, (out, _, _) = run(lib, 0, 1)
    assert rep == []
    assert (out[:128] == (np.arange(128) + 64) % 128 + 1).all()


def test_lane_exchange_without_a_wave_ordering_point(lib):
    rep, _ = run(lib, 1, 0)
    assert {(r[0], r[1]) for r in rep} == {("WAVE", "rk_lockstep")} and len(rep) == 2, rep   # both directions
    rep, _ = run(lib, 1, 1)
    assert rep == []


def test_write_after_read(lib):
    rep, _ = run(lib, 2, 0)
    assert {(r[0], r[3]) for r in rep if "-> store" in r[5]} == {("INTRA", line_of(r"buf\[t\] = v \+ 1;"))}, rep
    assert any(r[5].split("->")[0].strip().startswith("lds load") for r in rep)
    rep, _ = run(lib, 2, 1)
    assert rep == []


def test_ticket_hand_off_between_workgroups(lib):
    # the hand-off as the product writes it: silent, and the sum is right
    rep, (_, _, out) = run(lib, 3, 1)
    assert rep == [], rep
    assert (out[:128] == 6 * np.arange(128) + 15).all()
    # no ticket: the other workgroups' loads of workgroup 0's slice are unordered
    rep, _ = run(lib, 3, 0)
    assert {r[0] for r in rep} == {"INTER"} and all("global store" in r[5] for r in rep), rep
    # ticket taken before the slice is written: nothing was released
    rep, _ = run(lib, 3, 2)
    assert {r[0] for r in rep} == {"INTER"}, rep
    # ticket in place, but the reading waves do not wait for the work-item that took it
    rep, _ = run(lib, 3, 3)
    assert {r[0] for r in rep} == {"INTER"}, rep


def test_same_value_stores_are_told_apart(lib):
    rep, (out, _, _) = run(lib, 4)
    assert rep and all(r[5].startswith("same-value stores: ") for r in rep), rep
    assert {r[0] for r in rep} <= {"INTRA", "WAVE"}
    assert (out[:128] == 1).all()


def test_plain_load_of_an_accumulator_other_workgroups_add_to(lib):
    rep, _ = run(lib, 5)
    assert [(r[0], r[1]) for r in rep] == [("INTER", "rk_atomic_then_plain")], rep


def test_counter_read_by_every_lane_and_advanced_by_one(lib):
    """the latest reader of a word is not the only one that matters: lanes 0..62 read, lane 63 reads and stores"""
    rep, (out, _, _) = run(lib, 6, 0)
    assert [(r[0], r[1]) for r in rep] == [("WAVE", "rk_counter")] and "load" in rep[0][5].split("->")[0], rep
    assert (out[:64] == 5 + np.arange(64)).all()
    rep, _ = run(lib, 6, 1)
    assert rep == []


def test_uninitialised_lds_and_lds_beyond_the_block(lib):
    rep, _ = run(lib, 7, 0, uninit=True)
    assert [(r[0], r[2]) for r in rep] == [("UNINIT", line_of(r"= dyn\[t\];"))], rep
    rep, _ = run(lib, 7, 1, uninit=True)
    assert [(r[0], r[2]) for r in rep] == [("LDSOOB", line_of(r"dyn\[64 \+ t\] = t;"))], rep
    # a struct stored as a whole reaches the detector as a memory intrinsic: seen as a store (no UNINIT, no race)
    rep, (out, _, _) = run(lib, 7, 2, uninit=True)
    assert rep == [], rep
    assert (out[:64] == (63 - np.arange(64)) + 4).all()


def test_partial_barrier_from_an_lds_counter(lib):
    """two of a workgroup's four waves exchange a buffer behind a barrier of their own -- arrivals counted by an LDS atomic,
    then a poll (gridgcn_bwdfused.hip: half_barrier).  Without it: INTRA, both directions; with it: ordered, silent, and
    the exchanged values are the partner's."""
    rep, (out, _, _) = run(lib, 9, 0)
    st, ld = line_of(r"buf\[t\] = t \+ 1;", 2), line_of(r"= buf\[\(t \+ 64\) % 128\];")
    assert {(r[0], r[1]) for r in rep} == {("INTRA", "rk_ldsbar")}, rep
    assert {(r[2], r[3]) for r in rep} <= {(st, ld), (ld, st)}, rep
    rep, (out, _, _) = run(lib, 9, 1)
    assert rep == [], rep
    assert (out[:128] == (np.arange(128) + 64) % 128 + 1).all()


def test_traffic_accounting_on_a_kernel_with_known_footprint(lib):
    """simt_traffic_enable: requested bytes, 128-byte lines fetched per XCD (workgroup w on XCD w mod 8; a line an XCD has
    written or read earlier in the launch is not fetched again), distinct 32-byte sectors written -- on a kernel whose
    numbers can be counted by hand (tests/simt/race_selftest.hip: rk_traffic).  The race build of the product reports
    the same quantities per kernel: tools/simt_traffic.py, profiles/r6_emulated_traffic.txt."""
    a = np.zeros(4096, np.int32)
    b = np.zeros(16 * 64 * 16, np.int32)
    c = np.zeros(4096, np.int32)
    # (128-byte aligned views: the line counts below assume it)
    def aligned(x, n):
        off = (-x.ctypes.data % 128) // 4
        return x[off:off + n]
    a, b, c = aligned(np.zeros(4096 + 32, np.int32), 4096), aligned(np.zeros(16 * 64 * 16 + 32, np.int32), 16 * 64 * 16), \
        aligned(np.zeros(4096 + 32, np.int32), 4096)
    a[:] = np.arange(4096)
    lib.simt_traffic_report.restype = ctypes.c_int
    lib.simt_traffic_report.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    lib.simt_race_enable(0)
    lib.simt_traffic_reset()
    lib.simt_traffic_enable(1)
    try:
        lib.rk_run(8, 0, a.ctypes.data, b.ctypes.data, c.ctypes.data)
        buf = ctypes.create_string_buffer(1 << 16)
        lib.simt_traffic_report(buf, len(buf), 0)
    finally:
        lib.simt_traffic_enable(0)
        lib.simt_race_enable(1)
    rows = [ln.split("\t") for ln in buf.value.decode().splitlines() if ln.startswith("K")]
    assert len(rows) == 1 and rows[0][1] == "rk_traffic", rows
    launches, wgs, req_ld, req_st, fetched, written = map(int, rows[0][2:8])
    assert (launches, wgs) == (1, 16)
    assert req_ld == 16 * 64 * 4 * 2 + 8 * 64 * 4            # table + a by everyone, c by the upper eight workgroups
    assert req_st == 16 * 64 * 4 + 8 * 64 * 4                # b by everyone, c by the lower eight
    assert fetched == 8 * 256 + 16 * 256                     # the table once per XCD; a streamed; c never (written there)
    assert written == 16 * 64 * 32 + 8 * 64 * 4              # b: a sector per work-item; c: 8 x 256 contiguous bytes
    v = np.arange(2048, 2048 + 64)
    assert (b[::16][:64] == a[:64] + v).all() and (b[::16][8 * 64:9 * 64] == a[8 * 64:9 * 64] + v + a[:64] + v).all()
