// gridgcn_gemm.hip -- the small dense products beside the edge pipeline, fp32 MFMA (gfx950).
//
// GridConv's first point conv is applied to the SOURCE points and gathered (gridgcn_edgelin.hip), which
// leaves three small GEMMs per layer outside the per-edge kernels (one conv over the gathered [B,O,P,C]
// tensor in the reference: segmentation/models/gcn_module_g_att.py:120-170):
//   Ysrc  [R, C0] = feat [R, Cf] * W0f^T             mode 0 (both operands row-major along k)
//   gsrc  [R, Cf] = dYsrc [R, C0] * W0f              mode 1
//   dW0f  [C0, Cf] = dYsrc^T * feat,  Ysrc^T * Gsum  mode 2 (contraction over the R rows)
// with R = B * Nsrc <= a few ten thousand rows, Cf, C0 <= 512, operands that are column slices of wider
// tensors (row strides 4 + Cf, rot + Cf; not 16-byte aligned).  rocBLAS spent 21 launches and ~215 us of
// a cfg4 step on them (a batched GEMM over 128-row slabs + a sum for every transposed product).
// Here the operands are read straight into the MFMA fragment layout (no packing, any row stride):
//   modes 0/1: one wave per 32 x 32 output tile, K in chunks of 8 with FOUR chunks' loads in flight;
//   mode 2:    the K rows are cut into slices of GG_TN_ROWS over the workgroups (x) of a tile (y); the
//              four waves of a workgroup add their tiles in LDS; the workgroup that draws the last
//              ticket of its tile adds the slices in the order 0..S-1 (bit-reproducible) and hands the
//              ticket back as 0 -- one launch, device-scope stores/loads instead of a fence
//              (gg_k_dw_reduce_direct's hand-off).
#include "gridgcn_mma.h"

struct GGGemm {
    const float *A, *B;
    float *C;
    int M, N, K, lda, ldb, ldc;
    int zero_left;        // mode 1 only: columns [-zero_left, 0) left of C are zero-filled
    const float *bias;    // modes 0 / 1: added per output column (nullptr: none)
    float *part;          // mode 2: [tiles][S][1024]
    int *tick;            // mode 2: [tiles], zero on entry, zero again on exit
    int rps;              // mode 2: rows of one workgroup's slice (a multiple of GG_TN_ROWS)
};

// mode 0: C[m][n] = sum_k A[m][k] B[n][k]      mode 1: C[m][n] = sum_k A[m][k] B[k][n]
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gg_k_gemm_rows(GGGemm p)
{
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int ntn = (p.N + 31) >> 5;
    const int tile = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int ntile = ((p.M + 31) >> 5) * ntn;
    if (tile >= ntile) return;
    const int tm = tile / ntn, tn = tile - tm * ntn;
    int row = tm * 32 + l31, col = tn * 32 + l31;
    const bool colok = col < p.N;
    if (row >= p.M) row = p.M - 1;
    if (!colok) col = p.N - 1;
    // k in chunks of 8: the low half-wave takes k = kb .. kb+3, the high one kb+4 .. kb+7 (both operands)
    const float *ar = p.A + (size_t)row * p.lda + 4 * h;
    const float *br = MODE == 0 ? p.B + (size_t)col * p.ldb + 4 * h : p.B + col + (size_t)(4 * h) * p.ldb;
    const size_t bs = MODE == 0 ? 1 : (size_t)p.ldb;        // k stride of the B operand
    ggm_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    constexpr int U = 4;                                       // chunks in flight
    // (16-byte loads of the A rows pay in mode 1 only: measured 27 -> 13 us there, 7-15 -> 15-32 us in mode 0)
    const bool a16 = MODE == 1 && ((p.lda & 3) == 0) && (((size_t)p.A & 15) == 0);   // (wave uniform)
    int kb = 0;
    for (; kb + 8 * U <= p.K; kb += 8 * U) {
        float a[U][4], b[U][4];
        if (a16) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const float4 t = *(const float4 *)(ar + kb + 8 * u);
                a[u][0] = t.x; a[u][1] = t.y; a[u][2] = t.z; a[u][3] = t.w;
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int i = 0; i < 4; i++) a[u][i] = ar[kb + 8 * u + i];
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int i = 0; i < 4; i++) b[u][i] = br[(size_t)(kb + 8 * u + i) * bs];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i], b[u][i], acc, 0, 0, 0);
    }
    for (; kb < p.K; kb += 8) {
        // (the last chunk may be short -- K need not be a multiple of 8: zeros beyond K, loads from valid addresses)
        float a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool ok = kb + 4 * h + i < p.K;
            const float av = *(ok ? ar + kb + i : p.A), bv = *(ok ? br + (size_t)(kb + i) * bs : p.B);
            a[i] = ok ? av : 0.f; b[i] = ok ? bv : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc, 0, 0, 0);
    }
    // D layout: lane = column, register r = row (r & 3) + 8 (r >> 2) + 4 h
    float *cp = p.C + (size_t)(tm * 32 + 4 * h) * p.ldc + tn * 32 + l31;
    const float bias = (p.bias && colok) ? p.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int rr = (r & 3) + 8 * (r >> 2);
        if (colok && tm * 32 + 4 * h + rr < p.M) cp[(size_t)rr * p.ldc] = acc[r] + bias;
    }
    if (MODE == 1 && p.zero_left && tn == 0 && l31 < p.zero_left) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int rr = (r & 3) + 8 * (r >> 2);
            if (tm * 32 + 4 * h + rr < p.M) p.C[(long long)(tm * 32 + 4 * h + rr) * p.ldc - p.zero_left + l31] = 0.f;   // (columns LEFT of C: a signed offset)
        }
    }
}

#define GG_TN_ROWS 256      // rows of a workgroup's slice (64 per wave: 32 MFMA steps) ...
#define GG_TN_MAX_PART 4096 // ... or a multiple of that, so that tiles x slices <= this many 4-KB partial tiles (16 MB):
                            // a [512 x 320] product over 10^6 rows would otherwise ask for 2.6 GB of workspace

// C[m][n] = sum_k A[k][m] B[k][n]
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gg_k_gemm_tn(GGGemm p)
{
    __shared__ float red[3][16 * 64];
    __shared__ int s_last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
    const int ntn = (p.N + 31) >> 5;
    const int tile = blockIdx.y, tm = tile / ntn, tn = tile - tm * ntn;
    const int S = gridDim.x, z = blockIdx.x;
    int m = tm * 32 + l31, n = tn * 32 + l31;
    if (m >= p.M) m = p.M - 1;
    if (n >= p.N) n = p.N - 1;
    const long long ka = (long long)z * p.rps + wave * (p.rps / 4);
    long long kz = ka + p.rps / 4;
    if (kz > p.K) kz = p.K;
    ggm_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    const float *ap = p.A + m, *bp = p.B + n;
    long long k = ka + h;                                          // this half-wave's row of a step
    // sixteen, then eight steps (32 / 16 rows) with their loads issued together; uniform trip counts (MFMAs need
    // every lane).  (A wave's 64 rows were four rounds of eight steps: four memory latencies in a 20-us kernel.)
    for (; (k - h) + 31 < kz; k += 32) {
        float a[16], b[16];
#pragma unroll
        for (int i = 0; i < 16; i++) { a[i] = ap[(size_t)(k + 2 * i) * p.lda]; b[i] = bp[(size_t)(k + 2 * i) * p.ldb]; }
#pragma unroll
        for (int i = 0; i < 16; i++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc, 0, 0, 0);
    }
    for (; (k - h) + 15 < kz; k += 16) {
        float a[8], b[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { a[i] = ap[(size_t)(k + 2 * i) * p.lda]; b[i] = bp[(size_t)(k + 2 * i) * p.ldb]; }
#pragma unroll
        for (int i = 0; i < 8; i++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc, 0, 0, 0);
    }
    for (; k - h < kz; k += 2) {
        const bool ok = k < kz;
        const float a = ok ? ap[(size_t)k * p.lda] : 0.f, b = ok ? bp[(size_t)k * p.ldb] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    // the four waves' tiles, added in the order 0..3
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) red[wave - 1][r * 64 + lane] = acc[r];
    }
    __syncthreads();
    float *slot = p.part + ((size_t)tile * S + z) * 1024;
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float s = acc[r];
            s += red[0][r * 64 + lane];
            s += red[1][r * 64 + lane];
            s += red[2][r * 64 + lane];
            acc[r] = s;
            if (S > 1) __hip_atomic_store(&slot[r * 64 + lane], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (S > 1) {
        // Hand-off partials -> ticket -> last arriver.  The partials are written through (sc1 stores) and
        // drained (vmcnt(0)) before the barrier, which is what gfx950 needs; the release on the ticket
        // and the acquire fence of the last arriver are what the HIP memory model asks for on top
        // (one thread per workgroup pays: the release finds nothing dirty to write back).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const int t = __hip_atomic_fetch_add(&p.tick[tile], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            s_last = t == S - 1;
            if (s_last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(&p.tick[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        if (!s_last) return;
    }
    // the last arriver: slices 0..S-1 in order; 256 threads x 4 CONSECUTIVE elements of the tile, one 16-byte
    // load per slice and eight slices in flight (behind the acquire above plain loads are valid: MI355X_MICROARCH,
    // "consumer: one agent acquire -> __syncthreads() -> plain loads").  One element at a time -- 4 x 8 rounds of
    // dword loads for 64 slices -- was 16 of this kernel's 28 us at 16 K rows.
    if (S > 1) {
        const int e = threadIdx.x * 4;
        const float *q = p.part + (size_t)tile * S * 1024 + e;
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
        int zz = 0;
        for (; zz + 8 <= S; zz += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = *(const float4 *)(q + (size_t)(zz + u) * 1024);
#pragma unroll
            for (int u = 0; u < 8; u++) { s4.x += v[u].x; s4.y += v[u].y; s4.z += v[u].z; s4.w += v[u].w; }
        }
        for (; zz < S; zz++) {
            const float4 v = *(const float4 *)(q + (size_t)zz * 1024);
            s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
        }
        const int r = e >> 6, ln = e & 63;
        const int row = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), col = tn * 32 + (ln & 31);
        const float sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (row < p.M && col + i < p.N) p.C[(size_t)row * p.ldc + col + i] = sv[i];
    }
    if (S == 1 && wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, col = tn * 32 + l31;
            if (row < p.M && col < p.N) p.C[(size_t)row * p.ldc + col] = acc[r];
        }
    }
}

// rows per slice and number of slices of the transposed product
static void gg_tn_slices(int M, int N, int K, int &rps, int &S)
{
    const long long tiles = (long long)((M + 31) / 32) * ((N + 31) / 32);
    const long long s0 = ((long long)K + GG_TN_ROWS - 1) / GG_TN_ROWS;
    long long smax = GG_TN_MAX_PART / tiles;
    if (smax < 1) smax = 1;
    const long long nsub = (s0 + smax - 1) / smax;            // sub-slices of GG_TN_ROWS rows a workgroup walks
    rps = (int)(nsub * GG_TN_ROWS);
    S = (int)(((long long)K + rps - 1) / rps);
}

size_t gg_gemm_small_workspace(int M, int N, int K)
{
    const size_t tiles = (size_t)((M + 31) / 32) * ((N + 31) / 32);
    int rps, S;
    gg_tn_slices(M, N, K, rps, S);
    return tiles * (size_t)S * 1024 * sizeof(float) + tiles * sizeof(int) + 256;
}

// workspace (mode 2 only): gg_gemm_small_workspace bytes, its LAST tiles * 4 + 256 bytes (the tickets) zero
// on first use -- the kernel leaves them zero
int gg_gemm_small(int mode, const float *A, int lda, const float *B, int ldb, float *C, int ldc, int M, int N,
                  int K, int zero_left, const float *bias, void *ws, hipStream_t st)
{
    if (!A || !B || !C || M < 1 || N < 1 || K < 1 || mode < 0 || mode > 2) return 1;
    GGGemm p;
    p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.zero_left = mode == 1 ? zero_left : 0;
    p.bias = mode == 2 ? nullptr : bias;
    p.part = nullptr; p.tick = nullptr;
    const int ntile = ((M + 31) / 32) * ((N + 31) / 32);
    if (mode == 2) {
        if (zero_left || !ws) return 1;
        int S;
        gg_tn_slices(M, N, K, p.rps, S);
        p.part = (float *)ws;
        p.tick = (int *)((char *)ws + (size_t)ntile * S * 1024 * sizeof(float));
        gg_k_gemm_tn<<<dim3(S, ntile), 256, 0, st>>>(p);
    } else {
        if (zero_left < 0 || zero_left > 32) return 1;
        if (mode == 0) gg_k_gemm_rows<0><<<(ntile + 3) / 4, 256, 0, st>>>(p);
        else gg_k_gemm_rows<1><<<(ntile + 3) / 4, 256, 0, st>>>(p);
    }
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
