// Do HBM streaming and the fp32 MFMA pipe overlap on gfx950, and under which kernel structure?
// (the round-3 form of this file ran ONE wave per SIMD with 7 KB in flight and read
// "the times add up" off a latency-bound wave.)  A sweep over structure x occupancy x bytes in flight:
//
//   per "step" a compute wave issues 16 v_mfma_f32_32x32x2_f32 (65 536 flop) and NL KB are streamed for it
//   (NL = 2: 32 flop/B, the dW kernels; 3: 21.3 flop/B, about the ridge 157.3 TF / 8 TB/s = 19.7 -- the fused
//   attention backward sits at 18.3; 4: 16 flop/B)
//
//   mode A  loads into registers, consumed by VALU adds in the MFMA wave itself (what the round-3 kernels do)
//   mode B  global_load_lds_dwordx4 into a per-wave LDS ring, read back (ds_read_b128) by the same wave
//   mode C  dedicated loader waves next to MFMA-only waves, NO data dependence between them (4 + 4 or
//           4 + 12 waves per CU): the upper bound of any loader / consumer split
//   mode D  loader waves fill a two-half LDS ring with global_load_lds, the MFMA waves read their operands from
//           it, one workgroup barrier per half (the practical form of C)
//
// Reported per cell: ms, TFLOP/s (of 157.3), TB/s (of the 6.3 a pure copy reaches and of the 8.0 on the data
// sheet).  "n/a" = the cell does not fit (registers or LDS).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_mem.hip -o /tmp/mfma_mem && /tmp/mfma_mem
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))

// LDS reads that the compiler cannot see (it would put s_waitcnt vmcnt(0) in front of every ds_read that
// may alias an outstanding LDS-DMA write, i.e. drain the ring it is supposed to keep in flight)
__device__ __forceinline__ f4v lds_rd128(const float *p)
{
    f4v r;
    const unsigned a = (unsigned)(size_t)(LDS_AS const float *)p;
    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(a));
    return r;
}
#define LDS_WAIT(q, n) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
                            for (int l_ = 0; l_ < (n); l_++) asm volatile("" : "+v"(q[l_])); } while (0)

// Plain loads, kept in flight by construction: the group of loads for step s + DS - 1 is issued, a scheduling
// barrier pins it there, and the values of step s pass through an opaque "+v" asm right where they are consumed
// -- without it the compiler hoists the CONSUMING adds up to the load (round 3's form of this file: the ring
// then holds sums, every load is waited for at once and the loop measures latency); the compiler's own
// s_waitcnt vmcnt(N) then has N = the loads issued since (checked in the ISA of every cell: tools/micro/check_waits.py).
#define PIN(x) asm volatile("" : "+v"(x))

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---- mode A: register ring, VALU consumption in the MFMA wave.  RK = KB in flight per wave ----
template <int NL, int RK, int WPS>
__global__ __launch_bounds__(256, WPS) void k_a(const float4 *src, size_t per_wave4, float *out, int iters)
{
    constexpr int DS = RK / NL;                   // steps in flight
    f16v acc[4];
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const float4 *p = src + wave * per_wave4 + lane;
    f4v ring[DS][NL];
#pragma unroll
    for (int d = 0; d < DS - 1; d++)
#pragma unroll
        for (int l = 0; l < NL; l++) ring[d][l] = *(const f4v *)(p + (size_t)(d * NL + l) * 64);
    float a = 1.f, b = 2.f;
    // (UR revolutions of the ring per loop body: at the loop header the compiler's wait insertion cannot tell
    //  the pending loads apart and drains the ring once -- vmcnt(NL) instead of vmcnt(NL * (DS - 1)); with UR = 4
    //  that costs one exposed latency per 4 * DS steps instead of every DS)
    constexpr int UR = 4;
    for (int it = 0; it < iters; it += UR * DS) {
#pragma unroll
        for (int dd = 0; dd < UR * DS; dd++) {
            const int d = dd % DS;
#pragma unroll
            for (int l = 0; l < NL; l++)
                ring[(d + DS - 1) % DS][l] = *(const f4v *)(p + (size_t)((it + dd + DS - 1) * NL + l) * 64);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int l = 0; l < NL; l++) PIN(ring[d][l]);
            a = ring[d][0].x; b = ring[d][NL - 1].w;
#pragma unroll
            for (int l = 0; l < NL; l++) a += ring[d][l].y + ring[d][l].z + ring[d][l].w;
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- mode B: per-wave LDS ring filled by global_load_lds_dwordx4 (1 KB per instruction), same wave reads ----
template <int NL, int RK, int WPS>
__global__ __launch_bounds__(256, WPS) void k_b(const float4 *src, size_t per_wave4, float *out, int iters)
{
    constexpr int DS = RK / NL;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f16v acc[4];
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t wave = (size_t)blockIdx.x * 4 + wv;
    const float4 *p = src + wave * per_wave4 + lane;
    float *ring = lds + (size_t)wv * RK * 256;     // RK KB per wave
    auto issue = [&](int step) {
#pragma unroll
        for (int l = 0; l < NL; l++)
            __builtin_amdgcn_global_load_lds((const GLB_AS void *)(p + (size_t)(step * NL + l) * 64),
                                             (LDS_AS void *)(ring + ((step % DS) * NL + l) * 256), 16, 0, 0);
    };
    for (int d = 0; d < DS - 1; d++) issue(d);
    float a = 1.f, b = 2.f;
    for (int it = 0; it < iters; it++) {
        issue(it + DS - 1);
        // the NL loads of step `it` have landed when at most NL * (DS - 1) newer ones are outstanding
        __builtin_amdgcn_s_waitcnt(0x0f70 | ((NL * (DS - 1)) & 15) | ((((NL * (DS - 1)) >> 4) & 3) << 14));
        const float *slot = ring + ((it % DS) * NL) * 256;
        f4v q[NL];
#pragma unroll
        for (int l = 0; l < NL; l++) q[l] = lds_rd128(slot + (l * 64 + lane) * 4);
        LDS_WAIT(q, NL);
        a = q[0].x; b = q[NL - 1].w;
#pragma unroll
        for (int l = 0; l < NL; l++) a += q[l].y + q[l].z;
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i & 3], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- mode C: 4 MFMA-only waves + LW loader waves per workgroup (one workgroup per CU), independent ----
// every loader streams (4 * NL / LW) KB per step with RK KB in flight and consumes it with VALU adds
template <int NL, int RK, int LW>
__global__ __launch_bounds__(64 * (4 + LW), 1) void k_c(const float4 *src, size_t per_wave4, float *out, int iters)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (wv < 4) {
        f16v acc[4];
        for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
        float a = 1.f + lane, b = 2.f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i & 3], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
        return;
    }
    // loader: nload = iters * 4 * NL / LW loads of 1 KB, RK in flight
    const size_t lw = (size_t)blockIdx.x * LW + (wv - 4);
    const float4 *p = src + lw * per_wave4 + lane;
    const int nload = iters * 4 * NL / LW;
    f4v ring[RK];
#pragma unroll
    for (int d = 0; d < RK - 1; d++) ring[d] = *(const f4v *)(p + (size_t)d * 64);
    float s = 0.f;
    constexpr int UR = 4;
    for (int it = 0; it < nload; it += UR * RK) {
#pragma unroll
        for (int dd = 0; dd < UR * RK; dd++) {
            const int d = dd % RK;
            ring[(d + RK - 1) % RK] = *(const f4v *)(p + (size_t)(it + dd + RK - 1) * 64);
            __builtin_amdgcn_sched_barrier(0);
            PIN(ring[d]);
            s += ring[d].x + ring[d].w;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- mode D: LW loader waves fill a two-half LDS ring (global_load_lds), the 4 MFMA waves read it ----
// half = HS steps of 4 * NL KB; one __syncthreads per half; ring = 2 * HS * 4 * NL KB <= 128 KB
template <int NL, int HS, int LW>
__global__ __launch_bounds__(64 * (4 + LW), 1) void k_d(const float4 *src, size_t per_cu4, float *out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int HK = HS * 4 * NL;               // KB per half
    const float4 *base = src + (size_t)blockIdx.x * per_cu4;
    f16v acc[4];
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    float a = 1.f, b = 2.f;
    const int nhalf = iters / HS;
    auto fill = [&](int h) {                       // loader waves: the KB slices of half h, strided over LW
        if (wv >= 4 && h < nhalf) {
            float *dst = lds + (size_t)(h & 1) * HK * 256;
            for (int kb = wv - 4; kb < HK; kb += LW)
                __builtin_amdgcn_global_load_lds((const GLB_AS void *)(base + ((size_t)h * HK + kb) * 64 + lane),
                                                 (LDS_AS void *)(dst + kb * 256), 16, 0, 0);
        }
    };
    fill(0);
    for (int h = 0; h < nhalf; h++) {
        if (wv >= 4) __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): half h has landed
        __syncthreads();
        fill(h + 1);
        if (wv < 4) {
            const float *half = lds + (size_t)(h & 1) * HK * 256;
#pragma unroll
            for (int st = 0; st < HS; st++) {
                f4v q[NL];
#pragma unroll
                for (int l = 0; l < NL; l++) q[l] = lds_rd128(half + (((st * 4 + wv) * NL + l) * 64 + lane) * 4);
                LDS_WAIT(q, NL);
                a = q[0].x; b = q[NL - 1].w;
#pragma unroll
                for (int l = 0; l < NL; l++) a += q[l].y + q[l].z;
#pragma unroll
                for (int i = 0; i < 16; i++) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i & 3], 0, 0, 0);
            }
        }
        // (the consumers are done with half h only at the NEXT barrier, and the loaders write half h + 1
        //  meanwhile: two halves suffice)
    }
    float s = 0.f;
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float4 *g_src;
static float *g_out;
static size_t g_src4;
static void report(const char *cell, float ms, double flop, double bytes)
{
    const double tf = flop / ms / 1e9, tb = bytes / ms / 1e9;
    printf("%-58s %7.3f ms  %6.1f TF (%4.2f)  %5.2f TB/s (%4.2f of 6.3, %4.2f of 8)\n", cell, ms, tf, tf / 157.3, tb,
           tb / 6.3, tb / 8.0);
}
template <class F> static float timeit(F launch)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipGetLastError());
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return best;
}

// total work is the same in every cell of a given NL: 1024 compute-wave-equivalents x ITERS steps
#define ITERS 1024
template <int NL, int RK, int WPS> static void run_a()
{
    char cell[128];
    snprintf(cell, sizeof cell, "A reg ring, VALU use   NL=%d %d wave/SIMD %2d KB/wave in flight", NL, WPS, RK);
    constexpr int regs = RK * 4 + 64 + 24;
    if (regs > 512 / WPS) { printf("%-58s n/a (registers)\n", cell); return; }
    const int nb = 256 * WPS, iters = ITERS * 4 / WPS;
    const size_t per_wave4 = (size_t)(iters + 5 * RK) * NL * 64;
    if (per_wave4 * nb * 4 > g_src4) { printf("%-58s n/a (buffer)\n", cell); return; }
    const float ms = timeit([&] { k_a<NL, RK, WPS><<<nb, 256>>>(g_src, per_wave4, g_out, iters); });
    report(cell, ms, (double)nb * 4 * iters * 16 * 4096.0, (double)nb * 4 * iters * NL * 1024.0);
}
template <int NL, int RK, int WPS> static void run_b()
{
    char cell[128];
    snprintf(cell, sizeof cell, "B LDS ring (load_lds)   NL=%d %d wave/SIMD %2d KB/wave in flight", NL, WPS, RK);
    const size_t lds = (size_t)4 * RK * 1024;
    if (lds * WPS > 160 * 1024) { printf("%-58s n/a (LDS)\n", cell); return; }
    const int nb = 256 * WPS, iters = ITERS * 4 / WPS;
    const size_t per_wave4 = (size_t)(iters + RK) * NL * 64;
    if (per_wave4 * nb * 4 > g_src4) { printf("%-58s n/a (buffer)\n", cell); return; }
    CHECK(hipFuncSetAttribute((const void *)k_b<NL, RK, WPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const float ms = timeit([&] { k_b<NL, RK, WPS><<<nb, 256, lds>>>(g_src, per_wave4, g_out, iters); });
    report(cell, ms, (double)nb * 4 * iters * 16 * 4096.0, (double)nb * 4 * iters * NL * 1024.0);
}
template <int NL, int RK, int LW> static void run_c()
{
    char cell[128];
    snprintf(cell, sizeof cell, "C 4 MFMA + %2d loader waves/CU, independent NL=%d %2d KB/loader", LW, NL, RK);
    const int nb = 256, iters = ITERS * 4;
    const size_t per_wave4 = ((size_t)iters * 4 * NL / LW + 5 * RK) * 64;
    if (per_wave4 * nb * LW > g_src4) { printf("%-58s n/a (buffer)\n", cell); return; }
    const float ms = timeit([&] { k_c<NL, RK, LW><<<nb, 64 * (4 + LW)>>>(g_src, per_wave4, g_out, iters); });
    report(cell, ms, (double)nb * 4 * iters * 16 * 4096.0, (double)nb * 4 * iters * NL * 1024.0);
}
template <int NL, int HS, int LW> static void run_d()
{
    char cell[128];
    snprintf(cell, sizeof cell, "D %2d loaders -> LDS halves of %3d KB -> 4 MFMA waves   NL=%d", LW, HS * 4 * NL, NL);
    const size_t lds = (size_t)2 * HS * 4 * NL * 1024;
    if (lds > 160 * 1024) { printf("%-58s n/a (LDS)\n", cell); return; }
    const int nb = 256, iters = ITERS * 4;
    const size_t per_cu4 = ((size_t)iters + HS) * 4 * NL * 64;
    if (per_cu4 * nb > g_src4) { printf("%-58s n/a (buffer)\n", cell); return; }
    CHECK(hipFuncSetAttribute((const void *)k_d<NL, HS, LW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const float ms = timeit([&] { k_d<NL, HS, LW><<<nb, 64 * (4 + LW), lds>>>(g_src, per_cu4, g_out, iters); });
    report(cell, ms, (double)nb * 4 * iters * 16 * 4096.0, (double)nb * 4 * iters * NL * 1024.0);
}
template <int NL> static void sweep()
{
    printf("\n== NL = %d KB per 16 MFMAs = %.1f flop/B (MFMA alone: %.3f ms at 157.3 TF; stream alone: %.3f ms at 6.3 TB/s)\n",
           NL, 65536.0 / (NL * 1024.0), 1024.0 * ITERS * 4 * 65536.0 / 157.3e9, 1024.0 * ITERS * 4 * NL * 1024.0 / 6.3e9);
    run_a<NL, 8, 1>(); run_a<NL, 16, 1>(); run_a<NL, 32, 1>();
    run_a<NL, 8, 2>(); run_a<NL, 16, 2>(); run_a<NL, 32, 2>();
    run_a<NL, 8, 4>(); run_a<NL, 16, 4>();
    run_b<NL, 8, 1>(); run_b<NL, 16, 1>(); run_b<NL, 32, 1>();
    run_b<NL, 8, 2>(); run_b<NL, 16, 2>();
    run_b<NL, 8, 4>();
    run_c<NL, 8, 4>(); run_c<NL, 16, 4>(); run_c<NL, 32, 4>();
    run_c<NL, 8, 12>(); run_c<NL, 16, 12>();
    run_d<NL, 2, 4>(); run_d<NL, 4, 4>(); run_d<NL, 2, 12>(); run_d<NL, 4, 12>();
}
int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    g_src4 = (size_t)20 << 26;                     // 20 Gi float4-bytes/16: 21.5 GB (a cell streams up to 17 GB)
    CHECK(hipMalloc(&g_src, g_src4 * sizeof(float4)));
    CHECK(hipMemset(g_src, 0, g_src4 * sizeof(float4)));
    CHECK(hipMalloc(&g_out, (size_t)1024 * 1024 * 4));
    // the two roofs alone, same loops
    printf("== roofs of these loops (work of one sweep cell)\n");
    {
        const int nb = 256, iters = ITERS * 4;
        float ms = timeit([&] { k_c<2, 8, 4><<<nb, 64 * 8>>>(g_src, 0, g_out, iters); });   // per_wave4 = 0: loaders hit L2
        report("MFMA waves + loaders re-reading one cached KB", ms, (double)nb * 4 * iters * 16 * 4096.0, 0.0);
    }
    sweep<2>();
    sweep<3>();
    sweep<4>();
    return 0;
}
