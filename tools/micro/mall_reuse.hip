// Does a consumer that walks a freshly written tensor BACKWARDS find its tail in the 256 MB memory-side
// cache?  Kernel W writes S MB front to back; kernel R then reads them front to back or back to front
// (and writes S/2 MB of its own, as a conv kernel does).  Reported: R's time and effective read rate.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mall_reuse.hip -o /tmp/mall && /tmp/mall
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void kw(float4 *dst, size_t n4, float v)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        dst[i] = make_float4(v, v + 1.f, v + 2.f, v + 3.f);
}
// persistent grid: block b takes chunk j = b, b + G, ... of 64 KB each, in forward or reverse chunk order
__global__ __launch_bounds__(256) void kr(const float4 *src, float4 *out, size_t nchunk, int reverse)
{
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t j = blockIdx.x; j < nchunk; j += gridDim.x) {
        const size_t c = reverse ? nchunk - 1 - j : j;
        const float4 *p = src + c * 4096;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const float4 t = p[u * 256 + threadIdx.x];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        acc.x += s.x; acc.y += s.y; acc.z += s.z; acc.w += s.w;
        // half as many bytes written as read
#pragma unroll
        for (int u = 0; u < 8; u++) out[c * 2048 + u * 256 + threadIdx.x] = s;
    }
    if (acc.x == 12345.f) out[0] = acc;
}
int main()
{
    const size_t maxb = (size_t)1400 << 20;
    float4 *a, *o;
    hipMalloc(&a, maxb); hipMalloc(&o, maxb / 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mb : {64, 128, 192, 256, 336, 512, 672, 1344}) {
        const size_t n4 = ((size_t)mb << 20) / 16, nchunk = n4 / 4096;
        for (int rev = 0; rev < 2; rev++) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                kw<<<2048, 256>>>(a, n4, (float)rep);
                hipEventRecord(e0);
                kr<<<1024, 256>>>(a, o, nchunk, rev);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            printf("%5d MB written, then read %s: %.3f ms  (%.2f TB/s read + %.2f TB/s written)\n", mb,
                   rev ? "back to front" : "front to back", best, mb / 1048.576 / best, mb / 2 / 1048.576 / best);
        }
    }
    return 0;
}
