// gridgcn_fixpt.h -- fp32 -> 64-bit fixed point for sums kept with INTEGER LDS atomics
// (gridgcn_edgelin.hip: ds_add_f32 costs ~2 cycles per lane on gfx950, ds_add_u64 an eighth of that).
// Plain arithmetic, also compiled for the host by tests/test_fixpt_host.py.
#pragma once
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define GG_FIX_FN __device__ __forceinline__
#else
#include <math.h>
#define GG_FIX_FN static inline
#endif

// scale 2^k with 2^39 <= m * 2^k < 2^40 for the (finite, non-negative) maximum m given by its bits,
// k kept within +-100 so that the scale and its inverse are fp32 numbers (m = 0: k = 100 -- whatever
// value then exceeds the headroom takes the side path, the rest is exact at that scale)
GG_FIX_FN int gg_fix_exp(unsigned mbits)
{
    const int e = (int)(mbits >> 23) - 127;          // floor(log2 m) for normal m (0 / denormal: -127)
    const int k = 39 - e;
    return k > 100 ? 100 : (k < -100 ? -100 : k);
}

// x = trunc(v) as a 64-bit integer for |v| < 2^47, v an fp32 number: 24 high bits and the exact
// remainder, both through the 32-bit converter (a double / int64 conversion is ~30 slow instructions)
GG_FIX_FN long long gg_fix_i64(float v)
{
    const float hf = truncf(v * 0x1p-24f);
    const float lf = __builtin_fmaf(-hf, 0x1p+24f, v);            // exact: the low bits of v
    return (long long)(int)hf * 16777216ll + (long long)(int)lf;
}
