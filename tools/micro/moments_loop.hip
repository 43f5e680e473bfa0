#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef int gg_rsrc __attribute__((ext_vector_type(4)));
__device__ float gg_buf_ld(gg_rsrc r, unsigned lane_bytes, unsigned uniform_bytes, int aux = 0) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ __forceinline__ gg_rsrc mk(const void *b){ unsigned long long a=(unsigned long long)b; gg_rsrc r; r.x=__builtin_amdgcn_readfirstlane((int)(unsigned)a); r.y=__builtin_amdgcn_readfirstlane((int)(unsigned)(a>>32)); r.z=-1; r.w=0x00020000; return r; }
// MODE 0: full; 1: no MFMA (loads + valu); 2: no loads (mfma + valu on registers); 3: full, B operand = different register
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k(const float *Z1, unsigned E, float sc, float sh, float *out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    const unsigned npair = E / 2;
    unsigned per = npair / nw; per &= ~15u;
    const unsigned lo = gw * per;
    const gg_rsrc rz = mk(Z1);
    f16v acc; for (int r = 0; r < 16; r++) acc[r] = 0.f;
    float s1f = 0.f;
    auto ld = [&](unsigned q, float (&v)[16]) {
        for (int i = 0; i < 16; i++) v[i] = MODE == 2 ? (float)(q + i) : gg_buf_ld(rz, q * 256u + lane * 4u + 256u * i, 0);
    };
    float v[16]; ld(lo, v);
    const unsigned nfull = per / 16;
    for (unsigned b = 0; b < nfull; b++) {
        const unsigned q = lo + 16 * b;
        float vn[16];
        if (b + 1 < nfull) ld(q + 16, vn);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const float a = fmaxf(fmaf(v[i], sc, sh), 0.f);
            s1f += a;
            if (MODE == 1) acc[i] += a * a;
            else if (MODE == 3) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s1f, acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc, 0, 0, 0);
        }
        if (b + 1 < nfull) for (int i = 0; i < 16; i++) v[i] = vn[i];
    }
    float t = s1f; for (int r = 0; r < 16; r++) t += acc[r];
    out[gw * 64 + lane] = t;
}
int main(int argc, char **argv)
{
    const unsigned E = 3276800; float *Z, *out;
    hipMalloc(&Z, (size_t)E * 128); hipMalloc(&out, 4096 * 4 * 64 * 4);
    hipMemset(Z, 0, (size_t)E * 128);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {256, 512, 1024}) for (int mode = 0; mode < 4; mode++) {
        float best = 1e9;
        for (int it = 0; it < 6; it++) {
            hipEventRecord(e0);
            if (mode == 0) k<0><<<grid, 256>>>(Z, E, 1.f, 0.1f, out);
            if (mode == 1) k<1><<<grid, 256>>>(Z, E, 1.f, 0.1f, out);
            if (mode == 2) k<2><<<grid, 256>>>(Z, E, 1.f, 0.1f, out);
            if (mode == 3) k<3><<<grid, 256>>>(Z, E, 1.f, 0.1f, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("grid %4d mode %d (%s): %.1f us\n", grid, mode, mode == 0 ? "full" : mode == 1 ? "no mfma" : mode == 2 ? "no loads" : "B != A", best * 1e3);
    }
    return 0;
}
