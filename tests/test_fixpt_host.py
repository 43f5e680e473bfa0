"""The fp32 -> 64-bit fixed-point arithmetic of the sparse edge_lin0 backward
(grid_gcn_amd/csrc/gridgcn_fixpt.h, used by gg_k_edge_lin0_bwd_sparse) compiled for the HOST and
checked exhaustively enough: the 24-high-bits + remainder split is an exact truncation for every
|v| < 2^47, and the scale exponent puts the maximum into [2^39, 2^40)."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
HDR = os.path.join(HERE, "..", "grid_gcn_amd", "csrc", "gridgcn_fixpt.h")

SRC = r"""
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include "gridgcn_fixpt.h"
static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static float f_of(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static uint32_t b_of(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }
int main(void)
{
    long bad = 0, n = 0;
    /* every exponent below 2^47, random mantissas and signs, plus the edges of each binade */
    for (int e = 0; e <= 127 + 46; e++) {
        for (int i = 0; i < 20000; i++) {
            uint32_t m = (i == 0) ? 0u : (i == 1) ? 0x7fffffu : (uint32_t)(rnd() & 0x7fffffu);
            uint32_t sign = (uint32_t)(rnd() & 1u) << 31;
            float v = f_of(sign | ((uint32_t)e << 23) | m);
            if (!(fabsf(v) < 0x1p+47f)) continue;
            long long want = (long long)truncl((long double)v);
            bad += gg_fix_i64(v) != want;
            n++;
        }
    }
    const float specials[] = {0.f, -0.f, 1.f, -1.f, 0.5f, -0.5f, 16777215.f, 16777216.f, 16777217.f, -16777216.f,
                              0x1p+46f, -0x1p+46f, 0x1.fffffep+46f, -0x1.fffffep+46f, 1e-30f, -1e-30f};
    for (unsigned i = 0; i < sizeof(specials) / sizeof(specials[0]); i++) {
        bad += gg_fix_i64(specials[i]) != (long long)truncl((long double)specials[i]);
        n++;
    }
    /* scale exponent: 2^39 <= m * 2^k < 2^40 wherever k is not clamped; clamped to +-100 elsewhere */
    long badk = 0, nk = 0;
    for (int i = 0; i < 2000000; i++) {
        uint32_t b = (uint32_t)rnd() & 0x7fffffffu;
        if (b >= 0x7f800000u) continue;
        int k = gg_fix_exp(b);
        if (k < -100 || k > 100) badk++;
        if (k > -100 && k < 100 && b >= 0x00800000u) {
            long double x = ldexpl((long double)f_of(b), k);
            badk += !(x >= 0x1p+39L && x < 0x1p+40L);
        }
        nk++;
    }
    badk += gg_fix_exp(0u) != 100;
    badk += gg_fix_exp(b_of(3.0e38f)) != 39 - 127;
    printf("%ld %ld %ld %ld\n", n, bad, nk, badk);
    return 0;
}
"""


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_fixed_point_split_is_exact_truncation(tmp_path):
    src = tmp_path / "t.c"
    src.write_text(SRC)
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-I", os.path.dirname(HDR), str(src), "-o", str(exe), "-lm"])
    n, bad, nk, badk = map(int, subprocess.check_output([str(exe)]).split())
    assert n > 3_000_000 and bad == 0
    assert nk > 1_900_000 and badk == 0
