// Does HBM streaming slow the MFMA pipe down?  Each wave runs 16 fp32 MFMAs per step and streams
// NL dwordx4 loads (1 KB per wave each) per step from its own region (8 steps in flight); reported: TFLOP/s, TB/s and the
// shader clock during the kernel (clock64 ticks per wall_clock64 tick, 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_mem.hip -o /tmp/mfma_mem && /tmp/mfma_mem
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NL, int NM>
__global__ __launch_bounds__(256) void k(const float4 *src, size_t per_wave4, float *out, int iters, long long *clk)
{
    const long long c0 = clock64(), w0 = wall_clock64();
    f16v acc[16];
    for (int i = 0; i < 16; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const float4 *p = src + wave * per_wave4 + lane;
    float4 ring[8][NL > 0 ? NL : 1];
    for (int d = 0; d < 7; d++)
        for (int l = 0; l < NL; l++) ring[d][l] = p[(size_t)(d * NL + l) * 64];
    float a = 1.f, b = 2.f;
    for (int it = 0; it < iters; it += 8) {
#pragma unroll
        for (int d = 0; d < 8; d++) {
#pragma unroll
            for (int l = 0; l < NL; l++) ring[(d + 7) & 7][l] = p[(size_t)((it + d + 7) * NL + l) * 64];
            if (NL > 0) {                       // every loaded quad is consumed (else the loads are dropped)
                a = ring[d][0].x; b = ring[d][NL - 1].w;
#pragma unroll
                for (int l = 0; l < NL; l++) a += ring[d][l].y + ring[d][l].z;
            }
#pragma unroll
            for (int i = 0; i < NM; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}
template <int NL, int NM>
static void run()
{
    const int nb = 256, iters = 2048;
    const size_t per_wave4 = (size_t)(iters + 16) * (NL > 0 ? NL : 1) * 64;
    float4 *src; float *out; long long *clk, h[2];
    hipMalloc(&src, per_wave4 * nb * 4 * sizeof(float4));
    hipMemset(src, 0, per_wave4 * nb * 4 * sizeof(float4));
    hipMalloc(&out, (size_t)nb * 256 * 4);
    hipMalloc(&clk, 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        k<NL, NM><<<nb, 256>>>(src, per_wave4, out, iters, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double fl = (double)nb * 4 * iters * NM * 4096.0, by = (double)nb * 4 * iters * NL * 1024.0;
    printf("%2d MFMA + %d x 1 KB loads per step: %.3f ms  %6.1f TFLOP/s  %5.2f TB/s  shader clock %.2f GHz\n", NM, NL, best,
           fl / best / 1e9, by / best / 1e9, 0.1 * (double)h[0] / (double)h[1]);
    hipFree(src); hipFree(out); hipFree(clk);
}
int main()
{
    run<0, 16>(); run<1, 16>(); run<2, 16>(); run<3, 16>(); run<4, 16>();
    run<2, 1>(); run<4, 1>();           // (one MFMA per step: the streaming rate of the same loop)
    return 0;
}
