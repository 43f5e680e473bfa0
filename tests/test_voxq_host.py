"""The voxel quotient without the IEEE divide (grid_gcn_amd/csrc/gridgcn_voxq.h, used by every index
kernel through gg_voxel_of) compiled for the HOST and compared with floorf(a / d) -- the reference's
arithmetic, gridify.cu:134-138 -- on adversarial inputs: quotients within a few ulp of every integer
up to 2^24, the shipped voxel sizes, random bit patterns, denormals, NaN, Inf."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
HDR = os.path.join(HERE, "..", "grid_gcn_amd", "csrc", "gridgcn_voxq.h")

SRC = r"""
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include "gridgcn_voxq.h"
static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static float f_of(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static uint32_t b_of(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }
static long n = 0, bad = 0, slow = 0;
static void chk(float a, float d)
{
    volatile float r = 1.0f / d;
    volatile float want = floorf(a / d);
    float got = gg_floor_quot(a, d, r);
    /* same value, same sign of zero, NaN <-> NaN */
    int same = (b_of(got) == b_of(want)) || (got != got && want != want);
    bad += !same;
    n++;
    { float q = a * r; float dist = fabsf(q - rintf(q)); slow += !(dist > fmaxf(fabsf(q) * 0x1p-21f, 0x1p-100f)); }
}
int main(void)
{
    const float ds[] = {0.05f, 0.133333f, 0.4f, 0.25f, 2.0f, 2.0f / 64, 2.0f / 32, 0.1f, 0.3f, 1.0f / 3, 0.7f,
                        1e-3f, 3.0f, 1.0f, 0.0625f, 0.0123f, 7.7f, 1e-20f, 1e20f, 1.17549435e-38f};
    for (unsigned di = 0; di < sizeof(ds) / sizeof(ds[0]); di++) {
        const float d = ds[di];
        /* a such that a / d is within +-6 ulp of an integer k: every k near the shipped grid sizes, then a
           stride through all integers below 2^24 */
        for (long k = -70; k < (1l << 24) + 70; k += (k < 5000 ? 1 : 997)) {
            float a0 = (float)((double)k * (double)d);
            uint32_t b = b_of(a0);
            for (int u = -6; u <= 6; u++) {
                float a = f_of(b + (uint32_t)u);
                if (a != a) continue;
                chk(a, d);
            }
        }
        /* random magnitudes */
        for (int i = 0; i < 400000; i++) chk(f_of((uint32_t)rnd()), d);
        const float sp[] = {0.f, -0.f, 1.4e-45f, -1.4e-45f, 1e-40f, -1e-40f, INFINITY, -INFINITY, NAN,
                            3.4e38f, -3.4e38f, 8388608.f, 16777216.f, -8388608.5f};
        for (unsigned i = 0; i < sizeof(sp) / sizeof(sp[0]); i++) chk(sp[i], d);
    }
    /* random divisors, coordinates like the loaders' (unit ball + shift 1) */
    for (int i = 0; i < 4000000; i++) {
        float d = 0.01f + (float)(rnd() & 0xffffff) * (2.0f / 16777216.0f);
        float a = (float)((double)(rnd() & 0xffffff) / 16777216.0 * 2.6 - 0.3);
        chk(a, d);
    }
    printf("%ld %ld %ld\n", n, bad, slow);
    return 0;
}
"""


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_floor_quot_equals_ieee_division(tmp_path):
    src = tmp_path / "t.c"
    src.write_text(SRC)
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-I", os.path.dirname(HDR),
                           str(src), "-o", str(exe), "-lm"])
    n, bad, slow = map(int, subprocess.check_output([str(exe)]).split())
    assert n > 15_000_000 and bad == 0
    assert slow > 1000          # the guarded branch was exercised
