// How much of the MFMA rate survives when each wave issues NV independent VALU instructions per MFMA
// (the question behind "VALU per MFMA" in profiles/r2_pmc_bwd_gemm.txt).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu.hip -o /tmp/mfma_valu && /tmp/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NV, int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    f16v acc[8];
    for (int i = 0; i < 8; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    typedef float v2 __attribute__((ext_vector_type(2)));
    typedef float v4 __attribute__((ext_vector_type(4)));
    __shared__ float ldsbuf[4096];
    ldsbuf[threadIdx.x] = a;
    __syncthreads();
    const unsigned ldsaddr = (threadIdx.x & 63) * 16;
    v2 ab = {a, b};
    unsigned sc = 0;
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            // operands of the MFMA come from the VALU chain of an EARLIER iteration (v[i], v[i+8])
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(KIND == 2 ? v[i] : a, KIND == 2 ? v[i + 8] : b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; j++) {
                const int q = (i * NV + j) & 15;
                if (KIND == 0 || KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q]) : "v"(a), "v"(b));
                else if (KIND == 1) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(*(unsigned long long *)&v[q & 14]) : "s"((unsigned long long)iters));
                else if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(v2 *)&v[q & 14]) : "v"(ab));
                else if (KIND == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(*(v4 *)&v[q & 12]) : "v"(ldsaddr));
                else if (KIND == 5) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc));
                else if (KIND == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[q]) : "v"(a));
                else if (KIND == 7) asm volatile("v_max_f32 %0, 0, %0" : "+v"(v[q]));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    for (int i = 0; i < 16; i++) s += v[i];
    s += (float)sc;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV, int KIND>
static void run(int wg_per_cu)
{
    float *out;
    const int nb = 256 * wg_per_cu, iters = 8000 / wg_per_cu;
    hipMalloc(&out, (size_t)nb * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NV, KIND><<<nb, 256>>>(out, iters / 10, 1.f, 2.f);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        k<NV, KIND><<<nb, 256>>>(out, iters, 1.f, 2.f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double fl = (double)nb * 4 * iters * 8 * 4096.0;
    printf("%s x%2d per MFMA, %d wave(s)/SIMD: %.3f ms  %.1f TFLOP/s\n", KIND == 1 ? "v_lshl_add_u64" : (KIND == 2 ? "v_fma_f32->A/B " : KIND == 3 ? "v_pk_fma_f32   " : KIND == 4 ? "ds_read_b128   " : KIND == 5 ? "s_add_u32      " : KIND == 6 ? "v_cndmask_b32  " : KIND == 7 ? "v_max_f32      " : "v_fma_f32      "), NV, wg_per_cu, best, fl / best / 1e9);
    hipFree(out);
}
int main()
{
    run<0, 0>(1); run<0, 0>(2);
    run<2, 0>(1); run<2, 0>(2);
    run<4, 0>(1); run<4, 0>(2);
    run<8, 0>(1); run<8, 0>(2);
    run<12, 0>(1); run<12, 0>(2);
    run<16, 0>(2);
    run<4, 1>(2); run<8, 1>(2);
    run<4, 2>(2); run<8, 2>(2);
    run<4, 3>(2); run<8, 3>(2);
    run<2, 4>(2); run<4, 4>(2);
    run<4, 6>(2); run<8, 6>(2);
    run<4, 7>(2); run<8, 7>(2);
    return 0;
}
