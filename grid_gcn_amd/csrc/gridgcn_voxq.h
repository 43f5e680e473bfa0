// gridgcn_voxq.h -- floor((x + shift) / voxel_size) of gridify.cu:134-138 without the IEEE divide.
//
// The reference computes floorf(RN(a / d)) per coordinate, a = RN(x + shift).  hipcc expands an IEEE
// fp32 division into ~35 instructions; three of them per point were a third of the instruction stream
// of the first index kernel.  Here: q' = RN(a * r) with r = RN(1 / d) formed once on the host.  Both q'
// and Q = RN(a / d) lie within 2^-22 |t| of the real quotient t, so floor(q') can differ from floor(Q)
// only if an integer lies within 2^-22 |q'| of q'.  The guard asks for four times that distance
// (2^-21 |q'|) and otherwise -- also for NaN / Inf / quotients below 2^-100, where rounding is absolute --
// takes the exact division.  The result is therefore floorf(RN(a / d)) bit for bit, for every input;
// about 1 coordinate in 30 000 takes the slow branch.  Plain arithmetic: compiled for the host as well
// and checked against the division there (tests/test_voxq_host.py).
#pragma once
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define GG_VOXQ_FN __device__ __forceinline__
#define GG_VOXQ_MUL(a, b) __fmul_rn((a), (b))
#define GG_VOXQ_DIV(a, b) __fdiv_rn((a), (b))
#else
#include <math.h>
#define GG_VOXQ_FN static inline
#define GG_VOXQ_MUL(a, b) ((a) * (b))     /* build with -ffp-contract=off */
#define GG_VOXQ_DIV(a, b) ((a) / (b))
#endif

// floorf(a / d) with r = 1.0f / d (IEEE, formed by the caller once per call)
GG_VOXQ_FN float gg_floor_quot(float a, float d, float r)
{
    const float q = GG_VOXQ_MUL(a, r);
    const float dist = fabsf(q - rintf(q));
    const float lim = fmaxf(fabsf(q) * 0x1p-21f, 0x1p-100f);
    if (dist > lim) return floorf(q);          // (false for NaN / Inf: exact path)
    return floorf(GG_VOXQ_DIV(a, d));
}
