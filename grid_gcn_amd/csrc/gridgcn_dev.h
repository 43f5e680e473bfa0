// gridgcn_dev.h -- device helpers shared by the gfx950 kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gridgcn_voxq.h"

#define GG_WAVE 64
#define GG_PMAX 128   // max_p_grid limit (reference: best[128] gridifyknn.cu:257; LDS slot arrays)
#define GG_KMAX 7     // kernel_size limit (classification/configs/configs.yaml:49 uses 7)
#define GG_K3MAX (GG_KMAX * GG_KMAX * GG_KMAX)

struct GGGrid {
    float shift[3];
    float vs[3];
    float rvs[3];  // 1.0f / vs[j] (IEEE, host): gg_floor_quot
    int g[3];
    int G;     // g0*g1*g2
    int gxy;   // g0*g1
    int P, O, k, k3, loc;
    unsigned long long seed;
    const unsigned long long *seed_dev;  // optional device scalar added to `seed` (graph replay)
};

// effective sampling seed of the call: seed (+ *seed_dev when a device scalar is given, so that a
// captured hipGraph draws a fresh sample at every replay)
__device__ __forceinline__ unsigned long long gg_seed(const GGGrid &gp)
{
    return gp.seed_dev ? gp.seed + *gp.seed_dev : gp.seed;
}

// cuRAND XORWOW curand_init(seed,0,0) + first curand_uniform (gridify.cu:149-150).
// Restated integer-for-integer; rocRAND's XORWOW uses different scramble constants.
__device__ __forceinline__ float gg_xorwow_uniform(unsigned long long seed)
{
    unsigned s0 = ((unsigned)seed) ^ 0xaad26b49u;
    unsigned s1 = ((unsigned)(seed >> 32)) ^ 0xf7dcefddu;
    unsigned t0 = 1099087573u * s0;
    unsigned t1 = 2591861531u * s1;
    unsigned d = 6615241u + t1 + t0;
    unsigned v0 = 123456789u + t0;
    unsigned v4 = 5783321u + t0;
    unsigned t = v0 ^ (v0 >> 2);
    v4 = (v4 ^ (v4 << 4)) ^ (t ^ (t << 1));
    d += 362437u;
    unsigned x = v4 + d;
    // x*2^-32 is exact, so mul+add rounds once, like the FMA nvcc would emit.
    return __fadd_rn(__fmul_rn((float)x, 2.3283064e-10f), 1.1641532e-10f);
}

// "ceilf(curand_uniform(&state) * n) - 1"  (gridify.cu:150,183,261; gridify_up.cu:163)
__device__ __forceinline__ int gg_reservoir_pick(unsigned long long seed, int n)
{
    float u = gg_xorwow_uniform(seed);
    return (int)(ceilf(__fmul_rn(u, (float)n)) - 1.0f);
}

// voxel of a point (gridify.cu:134-143); -1 = dropped.  add then divide, IEEE, no contraction: the
// floor of the IEEE quotient comes from gg_floor_quot (one multiply + a guarded exact fallback,
// gridgcn_voxq.h: bit for bit floorf(__fdiv_rn(a, vs)))
__device__ __forceinline__ int gg_voxel_of(float x, float y, float z, const GGGrid &gp, int *c3)
{
    float f0 = gg_floor_quot(__fadd_rn(x, gp.shift[0]), gp.vs[0], gp.rvs[0]);
    float f1 = gg_floor_quot(__fadd_rn(y, gp.shift[1]), gp.vs[1], gp.rvs[1]);
    float f2 = gg_floor_quot(__fadd_rn(z, gp.shift[2]), gp.vs[2], gp.rvs[2]);
    bool ok = (f0 >= 0.0f) && (f0 < (float)gp.g[0]) && (f1 >= 0.0f) && (f1 < (float)gp.g[1]) &&
              (f2 >= 0.0f) && (f2 < (float)gp.g[2]);
    if (!ok) return -1;
    int c0 = (int)f0, c1 = (int)f1, c2 = (int)f2;
    if (c3) { c3[0] = c0; c3[1] = c1; c3[2] = c2; }
    return c2 * gp.gxy + c1 * gp.g[0] + c0;
}

// Phase stamps for tools/prof_phases.py: only in the -DGG_PROF build (libgridgcn_hip_prof.so).
// gg_prof_buf[(kernel*4096 + workgroup%4096)*16 + k] = 100 MHz wall clock at stamp k.
#ifdef GG_PROF
static __device__ unsigned long long *gg_prof_buf = nullptr;
#define GG_STAMP(kid, wg, k)                                                                   \
    do {                                                                                       \
        if (threadIdx.x == 0 && gg_prof_buf) {                                                 \
            unsigned long long *gg_pb_ =                                                       \
                gg_prof_buf + ((size_t)(kid) * 4096 + ((unsigned)(wg) & 4095u)) * 16;          \
            gg_pb_[(k)] = wall_clock64();                                                      \
            if ((k) == 0) gg_pb_[14] = clock64();   /* shader clock at the first stamp ... */  \
            gg_pb_[15] = clock64();                 /* ... and at the last one so far */       \
        }                                                                                      \
    } while (0)
#define GG_PROF_SETTER(name)                                                                   \
    extern "C" int name(void *buf)                                                             \
    {                                                                                          \
        return hipMemcpyToSymbol(HIP_SYMBOL(gg_prof_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : 3; \
    }
#else
#define GG_STAMP(kid, wg, k) do {} while (0)
#define GG_PROF_SETTER(name)
#endif

// Lanes of a wave execute in lockstep and their LDS operations complete in order: what one lane has stored is there
// for another lane's next load without any instruction in between.  The host-side emulation of the kernels
// (tests/simt/) runs the lanes of a wave one after the other between cross-lane operations and needs a rendezvous at
// such a point; GG_LOCKSTEP() marks it -- a wave barrier there (GG_SIMT), NOTHING in the GPU build.
#ifndef GG_LOCKSTEP
#ifdef GG_SIMT
#define GG_LOCKSTEP() __builtin_amdgcn_wave_barrier()
#else
#define GG_LOCKSTEP() ((void)0)
#endif
#endif

__device__ __forceinline__ int gg_lane() { return (int)(threadIdx.x & 63); }

// Workgroup number -> (cloud, item of the cloud) for a 1-D launch of B * per_cloud workgroups.
// Workgroup w is dispatched to XCD w % 8 (observed, MI355X_MICROARCH "Workgroup dispatch"; only
// speed depends on it): with B a multiple of 8 every kernel of a call sends the workgroups of
// cloud b to XCD b % 8, so what one kernel of the call leaves in that XCD's 4 MB L2 (the cloud's
// split runs, sorted ids, voxel table: ~3 MB at 81920 points) is what the next one reads.
__device__ __forceinline__ void gg_cloud_item(int wg, int per_cloud, int B, int &b, int &item)
{
    if ((B & 7) == 0) {
        const int xcd = wg & 7, j = wg >> 3;
        b = (j / per_cloud) * 8 + xcd;
        item = j % per_cloud;
    } else {
        b = wg / per_cloud;
        item = wg % per_cloud;
    }
}

// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ int gg_wave_incl_scan(int v)
{
    int lane = gg_lane();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

__device__ __forceinline__ int gg_wave_sum(int v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__device__ __forceinline__ long long gg_wave_sum_ll(long long v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
