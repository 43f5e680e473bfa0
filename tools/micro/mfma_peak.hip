// Sustained fp32 / bf16 MFMA rate of the chip as it is clocked under that load (the number the
// per-kernel "MFMA busy" shares should be read against, next to the 2.4 GHz data-sheet peak).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short s8v __attribute__((ext_vector_type(8)));
template <int NACC, bool BF>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    f16v acc[NACC];
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    s8v a8, b8;
    for (int i = 0; i < 8; i++) { a8[i] = (short)(threadIdx.x + i); b8[i] = (short)(threadIdx.x * 3 + i); }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) {
            if (BF) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a8), __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b8), acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, bool BF>
static void run(const char *name, int wg_per_cu, int iters, double flop_per_mfma)
{
    float *out;
    const int nb = 256 * wg_per_cu;
    hipMalloc(&out, (size_t)nb * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, BF><<<nb, 256>>>(out, iters / 10, 1.f, 2.f);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        k<NACC, BF><<<nb, 256>>>(out, iters, 1.f, 2.f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double fl = (double)nb * 4 * iters * NACC * flop_per_mfma;
        printf("%-28s waves/SIMD %d  %.3f ms  %.1f TFLOP/s  (implied clock at 256 flop/clk/CU fp32: %.2f GHz)\n", name, wg_per_cu, ms,
               fl / ms / 1e9, BF ? fl / ms / 1e9 / (256 * 4096.0 / 1e3) : fl / ms / 1e9 / (256 * 256.0 / 1e3));
    }
    hipFree(out);
}
int main()
{
    run<8, false>("fp32 32x32x2, 8 acc", 1, 20000, 4096.0);
    run<8, false>("fp32 32x32x2, 8 acc", 2, 10000, 4096.0);
    run<4, false>("fp32 32x32x2, 4 acc", 2, 20000, 4096.0);
    run<8, true>("bf16 32x32x16, 8 acc", 1, 20000, 32768.0);
    run<8, true>("bf16 32x32x16, 8 acc", 2, 10000, 32768.0);
    return 0;
}
