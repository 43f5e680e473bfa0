// gridgcn_train.hip -- training-mode building blocks of the per-edge MLPs (gfx950, fp32 MFMA).
//
// The reference runs every 1x1 conv of sub_g_update as Convolution -> BatchNorm(batch statistics)
// -> ReLU (utils/ops.py:149-158), i.e. per layer a cuDNN GEMM, a statistics pass, a normalise pass
// and a ReLU pass over a [B,C,O,P] tensor, and the mirror image in backward.  Here, per layer:
//
//   gg_k_linear_fwd   Z = act(X)*W + b.  act() optionally applies the PREVIOUS layer's
//                     BatchNorm+ReLU while the tile is staged into LDS (normalised activations are
//                     never materialised); the epilogue accumulates sum / sum of squares of Z for
//                     THIS layer's BatchNorm.                                   [1 launch forward]
//   gg_k_linear_bwd   one pass over the edges: dZ = BN/ReLU backward of (dY, Z) computed while
//                     staging; dX = dZ * W^T and dW += act(Aprev)^T * dZ on MFMA from the same LDS
//                     tiles; the epilogue accumulates the BatchNorm-backward sums of the PREVIOUS
//                     layer from dX, so no separate reduce / element-wise passes remain
//                     (rocBLAS ran these tall-skinny GEMMs at ~13 TFLOP/s).      [1 launch backward]
//   gg_k_bn_apply / gg_k_bn_bwd_reduce   only at the two ends of an MLP.
//
// Both GEMM kernels are PERSISTENT: a workgroup loads the layer's (packed) weights into LDS once
// and then walks row tiles, so the B operand never waits on L2 (with 4-byte-per-lane global B
// loads the backward GEMM was latency bound: 10.7 ms for the 3.3 M-edge layer of cfg4).
#include "gridgcn_mma.h"
#include "gridgcn_once.h"
#include "gridgcn_train.h"

// kernel-selection option (include/gridgcn.h: gridgcn_set_option): the one-pass backward of the
// attention conv (gridgcn_attbwd.hip) on by default; 0 = the separate dX / dW kernels (A/B tests)
static int g_opt_att_bwd_fused = 1;
void gg_set_att_bwd_fused(int on) { g_opt_att_bwd_fused = on ? 1 : 0; }
int gg_get_att_bwd_fused() { return g_opt_att_bwd_fused; }

// ------------------------------------------------------------------------------------------
// copy a [nrows x cin] row-major chunk (contiguous in global memory) into an LDS tile with row
// stride ld, optionally through x -> relu(x*scale[c] + shift[c]); zero the K padding / missing rows.
template <bool XFORM>
__device__ __forceinline__ void gg_stage_rows(float *dst, int ld, const float *__restrict__ src,
                                              int nrows, int cin, int K,
                                              const float *__restrict__ scale,
                                              const float *__restrict__ shift, int tid, int nthr,
                                              int rt = 32)
{
    const int nel = nrows * cin;
    const float inv = 1.0f / (float)cin;
    // two chunks of 8 loads per thread are in flight at any time: the loads of chunk k+1 are
    // issued before chunk k is written to LDS, so only the first round trip to HBM is exposed
    constexpr int U = 8;
    const int step = nthr * U;
    float va[U], vb[U];
    auto ldg = [&](float (&v)[U], int base) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            int i = base + u * nthr + tid;
            v[u] = i < nel ? src[i] : 0.f;
        }
    };
    auto sts = [&](const float (&v)[U], int base) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            int i = base + u * nthr + tid;
            if (i < nel) {
                int r = (int)(((float)i + 0.5f) * inv);          // exact for nel <= 32768
                int c = i - r * cin;
                float x = v[u];
                if (XFORM) { x = x * scale[c] + shift[c]; x = x > 0.f ? x : 0.f; }
                dst[r * ld + c] = x;
            }
        }
    };
    ldg(va, 0);
    for (int base = 0; base < nel; base += 2 * step) {
        if (base + step < nel) ldg(vb, base + step);
        sts(va, base);
        if (base + 2 * step < nel) ldg(va, base + 2 * step);
        if (base + step < nel) sts(vb, base + step);
    }
    const int padc = K - cin;
    if (padc > 0)
        for (int i = tid; i < nrows * padc; i += nthr) {
            int r = i / padc, c = cin + (i - r * padc);
            dst[r * ld + c] = 0.f;
        }
    for (int i = nrows * K + tid; i < rt * K; i += nthr) {
        int r = i / K, c = i - r * K;
        dst[r * ld + c] = 0.f;
    }
}

// same, with 16-byte global loads: a tile of 32k rows starts 16-byte aligned for ANY row length
// (32 rows * 4 B), so the chunk is read as float4 even when cin is odd; elements are routed to
// their (row, column) one by one.  4-byte loads reach only about half the HBM rate on MI355X.
template <bool XFORM>
__device__ __forceinline__ void gg_stage_rows4(float *dst, int ld, const float *__restrict__ src,
                                               int nrows, int cin, int K,
                                               const float *__restrict__ scale,
                                               const float *__restrict__ shift, int tid, int nthr,
                                               int rt = 32)
{
    const int nel = nrows * cin;
    const int nq = (nel + 3) >> 2;
    const float inv = 1.0f / (float)cin;
    constexpr int U = 4;
    const int step = nthr * U;
    float4 va[U], vb[U];
    auto ldg = [&](float4 (&v)[U], int base) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int q = base + u * nthr + tid;
            const int i = q * 4;
            if (i + 3 < nel) v[u] = *(const float4 *)(src + i);
            else {
                v[u].x = i < nel ? src[i] : 0.f;
                v[u].y = i + 1 < nel ? src[i + 1] : 0.f;
                v[u].z = i + 2 < nel ? src[i + 2] : 0.f;
                v[u].w = 0.f;
            }
        }
    };
    auto sts = [&](const float4 (&v)[U], int base) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = (base + u * nthr + tid) * 4;
            if (i < nel) {
                int r = (int)(((float)i + 0.5f) * inv);
                int c = i - r * cin;
                const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (i + k < nel) {
                        float x = e[k];
                        if (XFORM) { x = x * scale[c] + shift[c]; x = x > 0.f ? x : 0.f; }
                        dst[r * ld + c] = x;
                    }
                    if (++c == cin) { c = 0; r++; }
                }
            }
        }
    };
    ldg(va, 0);
    for (int base = 0; base < nq; base += 2 * step) {
        if (base + step < nq) ldg(vb, base + step);
        sts(va, base);
        if (base + 2 * step < nq) ldg(va, base + 2 * step);
        if (base + step < nq) sts(vb, base + step);
    }
    const int padc = K - cin;
    if (padc > 0)
        for (int i = tid; i < nrows * padc; i += nthr) {
            int r = i / padc, c = cin + (i - r * padc);
            dst[r * ld + c] = 0.f;
        }
    for (int i = nrows * K + tid; i < rt * K; i += nthr) {
        int r = i / K, c = i - r * K;
        dst[r * ld + c] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// Register-resident tile: thread t of nthr holds float4 number (u*nthr + t) of a contiguous chunk
// of `nel` floats (a 32-row tile of a row-major matrix is one such chunk, 16-byte aligned for any
// row length because 32 rows * 4 B is a multiple of 16).  Loading a tile into registers BEFORE
// the MFMA phase of the previous tile and writing it to LDS AFTER lets the HBM round trip of
// tile t+1 hide behind the math of tile t (the phases were measured fully serialised:
// stage 1.0 ms + MFMA 0.65 ms + store 0.67 ms for one 3.3 M-row layer).
template <int NV> struct GGTileRegs { float4 v[NV]; };

template <int NV>
__device__ __forceinline__ void gg_tile_load(GGTileRegs<NV> &t, const float *__restrict__ src,
                                             int nel, int tid, int nthr)
{
#pragma unroll
    for (int u = 0; u < NV; u++) {
        const int i = (u * nthr + tid) * 4;
        if (i + 3 < nel) t.v[u] = *(const float4 *)(src + i);
        else {
            t.v[u].x = i < nel ? src[i] : 0.f;
            t.v[u].y = i + 1 < nel ? src[i + 1] : 0.f;
            t.v[u].z = i + 2 < nel ? src[i + 2] : 0.f;
            t.v[u].w = 0.f;
        }
    }
}

// calls f(r, c, value) for every element of the tile held in registers
template <int NV, class F>
__device__ __forceinline__ void gg_tile_foreach(const GGTileRegs<NV> &t, int nel, int cin, int tid,
                                                int nthr, F f)
{
    const float inv = 1.0f / (float)cin;
#pragma unroll
    for (int u = 0; u < NV; u++) {
        const int i = (u * nthr + tid) * 4;
        if (i < nel) {
            int r = (int)(((float)i + 0.5f) * inv);          // exact for nel <= 32768
            int c = i - r * cin;
            const float e[4] = {t.v[u].x, t.v[u].y, t.v[u].z, t.v[u].w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (i + k < nel) f(r, c, e[k], u * 4 + k);
                if (++c == cin) { c = 0; r++; }
            }
        }
    }
}

// zero the K padding columns and the rows beyond nrows of a [32][ld] LDS tile
__device__ __forceinline__ void gg_tile_pad(float *dst, int ld, int nrows, int cin, int K, int tid,
                                            int nthr)
{
    const int padc = K - cin;
    if (padc > 0)
        for (int i = tid; i < nrows * padc; i += nthr) {
            int r = i / padc, c = cin + (i - r * padc);
            dst[r * ld + c] = 0.f;
        }
    if (nrows < 32)
        for (int i = nrows * K + tid; i < 32 * K; i += nthr) {
            int r = i / K, c = i - r * K;
            dst[r * ld + c] = 0.f;
        }
}

__device__ __forceinline__ void gg_copy_to_lds(float *dst, const float *__restrict__ src, int n,
                                               int tid, int nthr)
{
    const int n4 = n >> 2;
    for (int i = tid; i < n4; i += nthr) ((float4 *)dst)[i] = ((const float4 *)src)[i];
    for (int i = (n4 << 2) + tid; i < n; i += nthr) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------
// forward: persistent workgroups of nw independent waves; each wave owns one 32-row tile at a
// time; WLDS: the packed weights live in LDS for the lifetime of the workgroup.
template <int NT, bool WLDS, int NV>
__global__ __launch_bounds__(256, 1) void gg_k_linear_fwd(GGLinFwd p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int lda = p.lda;
    float *Wl = lds;
    float *Aw = lds + (WLDS ? p.K * p.ldw : 0) + wave * 32 * lda;
    if (WLDS) {
        gg_copy_to_lds(Wl, p.W, p.K * p.ldw, tid, blockDim.x);
        __syncthreads();
    }
    static_assert(NV >= 1, "");
    const long long ntile = (p.E + 31) >> 5;
    const int ngroup = p.ldw / (32 * NT);
    float ssum[2][NT], ssq[2][NT];
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) { ssum[g][nt] = 0.f; ssq[g][nt] = 0.f; }

    // (a register-resident prefetch of the next tile -- gg_tile_load/gg_tile_foreach above -- was
    //  measured SLOWER here: the epilogue's 64 stores per lane sit behind the prefetch loads in the
    //  in-order vmcnt queue, so waiting for the loads also drains the stores.)
    for (long long tile = (long long)blockIdx.x * nw + wave; tile < ntile;
         tile += (long long)gridDim.x * nw) {
        const long long r0 = tile << 5;
        const int nrows = (p.E - r0 < 32) ? (int)(p.E - r0) : 32;
        if (p.scale) gg_stage_rows4<true>(Aw, lda, p.X + r0 * p.cin, nrows, p.cin, p.K, p.scale, p.shift, lane, 64);
        else gg_stage_rows4<false>(Aw, lda, p.X + r0 * p.cin, nrows, p.cin, p.K, nullptr, nullptr, lane, 64);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < 2; g++) {
            if (g >= ngroup) continue;
            ggm_f32x16 acc[NT];
            ggm_zero<NT>(acc);
            if (WLDS) ggm_mma_lds<NT>(Aw, lda, Wl + (size_t)p.K * g * 32 * NT, p.K, acc);
            else ggm_mma<NT>(Aw, lda, p.W + (size_t)p.K * g * 32 * NT, p.K, acc);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                const int col = (g * NT + nt) * 32 + (lane & 31);
                const bool cok = col < p.cout;
                const float bias = cok ? p.b[col] : 0.f;
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = ggm_row(r, lane);
                    const float z = acc[nt][r] + bias;
                    if (cok && row < nrows) {
                        p.Z[(r0 + row) * (p.ldz ? p.ldz : p.cout) + col] = z;
                        s += z;
                        q += z * z;
                    }
                }
                ssum[g][nt] += s;
                ssq[g][nt] += q;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            float s = ssum[g][nt] + __shfl_xor(ssum[g][nt], 32, 64);
            float q = ssq[g][nt] + __shfl_xor(ssq[g][nt], 32, 64);
            const int col = (g * NT + nt) * 32 + lane;
            if (lane < 32 && g < ngroup && col < p.cout) {
                atomicAdd(&p.sums[col], (double)s);
                atomicAdd(&p.sums[p.cout + col], (double)q);
            }
        }
}

template <int NT, int NV>
static int launch_fwd_nv(const GGLinFwd &q, hipStream_t st)
{
    static GGDevOnce attr_done;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)gg_k_linear_fwd<NT, true, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 3;
        if (hipFuncSetAttribute((const void *)gg_k_linear_fwd<NT, false, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 3;
        attr_done = true;
    }
    const long long ntile = (q.E + 31) >> 5;
    const size_t wbytes = (size_t)q.K * q.ldw * 4, abytes = (size_t)32 * q.lda * 4;
    const size_t cap = 158 * 1024;
    int nw = 4;
    bool wlds = true;
    if (wbytes + 4 * abytes > cap) {
        if (wbytes + 2 * abytes <= cap) nw = 2;
        else { wlds = false; nw = (4 * abytes <= cap) ? 4 : 1; if (abytes > cap) return 1; }
    }
    const size_t lds = (wlds ? wbytes : 0) + nw * abytes;
    const int per_cu = lds > 80 * 1024 ? 1 : 2;
    long long nb = (ntile + nw - 1) / nw;
    if (nb > 256 * per_cu) nb = 256 * per_cu;
    if (wlds) gg_k_linear_fwd<NT, true, NV><<<(int)nb, 64 * nw, lds, st>>>(q);
    else gg_k_linear_fwd<NT, false, NV><<<(int)nb, 64 * nw, lds, st>>>(q);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

template <int NT>
static int launch_fwd(const GGLinFwd &q, hipStream_t st)
{
    return launch_fwd_nv<NT, 1>(q, st);        // NV: register-tile size, unused by the shipped path
}

int gg_linear_fwd(const GGLinFwd &p, hipStream_t st)
{
    if (p.E < 1 || p.cin < 1 || p.K < 4 || (p.K & 3) || p.K < p.cin || p.cin > 1024) return 1;
    if (p.ldw != 32 && p.ldw != 64 && p.ldw != 128 && p.ldw != 256) return 1;
    GGLinFwd q = p;
    q.lda = p.K | 1;
    if (p.ldw == 32) return launch_fwd<1>(q, st);
    if (p.ldw == 64) return launch_fwd<2>(q, st);
    return launch_fwd<4>(q, st);
}

// ------------------------------------------------------------------------------------------
// backward of one (linear -> BatchNorm(batch stats) -> ReLU) layer.  256 threads = 4 waves share
// one 32-row tile; persistent workgroups (tile = blockIdx.x, += gridDim.x).
//   D  [32][ldd] = dZ tile        Zp [32][lda] = raw previous activation tile (Z_{l-1} or X)
//   GEMM1: dX[32 x cin] = D * Wb            column tiles listed in p.t1[wave]   (<= 3 per wave)
//   GEMM2: dW[cin x C] += act(Zp)^T * D     (m,n) tile pairs listed in p.t2[wave] (<= PAIRS)
// The host balances the two lists so that every wave issues the same number of MFMAs per tile.
template <int PAIRS, bool WLDS>
__global__ __launch_bounds__(256, 1) void gg_k_linear_bwd(GGLinBwd p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = p.C, cin = p.cin;
    const int ldd = p.ldd, lda = p.lda;
    const int C4 = (C + 3) & ~3;
    const int ntn1 = (cin + 31) >> 5;        // column tiles over cin
    const int ntn2 = (C + 31) >> 5;          // column tiles over C
    float *Wl = lds;                         // [ntn1][C4][32] when WLDS
    float *D = lds + (WLDS ? ntn1 * C4 * 32 : 0);   // [32][ldd]
    float *Zp = D + 32 * ldd;                // [32][lda]
    float *cst = Zp + 32 * lda;              // per-channel constants
    float *c_scale = cst, *c_shift = cst + C, *c_mean = cst + 2 * C, *c_rstd = cst + 3 * C;
    float *c_m1 = cst + 4 * C, *c_m2 = cst + 5 * C;
    float *c_ps = cst + 6 * C, *c_psh = c_ps + cin, *c_pm = c_psh + cin, *c_pr = c_pm + cin;
    int *s_am = (int *)(c_pr + cin);         // [ncen_max][C] arg max of the tile's centres
    float *s_gv = (float *)(s_am + p.ncen_max * C);
    for (int c = tid; c < C; c += 256) {
        c_scale[c] = p.scale[c]; c_shift[c] = p.shift[c]; c_mean[c] = p.mean[c];
        c_rstd[c] = p.rstd[c]; c_m1[c] = p.m1[c]; c_m2[c] = p.m2[c];
    }
    const bool prevbn = p.pscale != nullptr;
    for (int c = tid; c < cin; c += 256) {
        c_ps[c] = prevbn ? p.pscale[c] : 1.f; c_psh[c] = prevbn ? p.pshift[c] : 0.f;
        c_pm[c] = prevbn ? p.pmean[c] : 0.f; c_pr[c] = prevbn ? p.prstd[c] : 0.f;
    }
    if (WLDS && p.dX) gg_copy_to_lds(Wl, p.Wb, ntn1 * C4 * 32, tid, 256);
    // this wave's work lists (packed bytes, 0xff = none)
    const unsigned t1 = wave == 0 ? p.t1[0] : (wave == 1 ? p.t1[1] : (wave == 2 ? p.t1[2] : p.t1[3]));
    unsigned t2[3];
#pragma unroll
    for (int w = 0; w < 3; w++)
        t2[w] = wave == 0 ? p.t2[0][w] : (wave == 1 ? p.t2[1][w] : (wave == 2 ? p.t2[2][w] : p.t2[3][w]));
    const long long ntile = (p.E + 31) >> 5;

    ggm_f32x16 accW[PAIRS];
    ggm_zero<PAIRS>(accW);
    float s1[3] = {0.f, 0.f, 0.f}, s2[3] = {0.f, 0.f, 0.f};
    __syncthreads();

    for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const long long r0 = tile << 5;
        const int nrows = (p.E - r0 < 32) ? (int)(p.E - r0) : 32;
        // ---- sparse upstream gradient: (amax, gval) rows of the centres this tile touches ----
        int rem0 = 0;
        const float invP = 1.0f / (float)p.P;
        if (p.amax) {
            const long long o0 = r0 / p.P;
            rem0 = (int)(r0 - o0 * p.P);
            const int ncen = (rem0 + nrows - 1) / p.P + 1;
            const gg_amax_t *am = p.amax + o0 * C;
            const float *gv = p.gval + o0 * C;
            for (int i = tid; i < ncen * C; i += 256) { s_am[i] = am[i]; s_gv[i] = gv[i]; }
            __syncthreads();
        }
        // ---- stage D = dZ (BatchNorm+ReLU backward, element-wise part) ----
        {
            const float *dy = p.dY + r0 * C, *zz = p.Z + r0 * C;
            const int nel = nrows * C;
            const float inv = 1.0f / (float)C;
            // ping-pong chunks of 4+4 loads per thread: only the first HBM round trip is exposed
            constexpr int U = 4;
            float ga[U], za[U], gb[U], zb[U];
            auto ldg = [&](float (&g)[U], float (&z)[U], int base) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    int i = base + u * 256 + tid;
                    z[u] = i < nel ? zz[i] : 0.f;
                    g[u] = (!p.amax && i < nel) ? dy[i] : 0.f;
                }
            };
            auto sts = [&](const float (&g)[U], const float (&z)[U], int base) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    int i = base + u * 256 + tid;
                    if (i < nel) {
                        int r = (int)(((float)i + 0.5f) * inv);
                        int c = i - r * C;
                        float sc = c_scale[c];
                        float gg = g[u];
                        if (p.amax) {
                            // sparse upstream gradient (max over P): only the arg-max edge of a
                            // (centre, channel) carries gval; the tile's centres sit in LDS
                            int t = rem0 + r;
                            int oc = (int)(((float)t + 0.5f) * invP);
                            int pp = t - oc * p.P;
                            gg = (s_am[oc * C + c] == pp) ? s_gv[oc * C + c] : 0.f;
                        }
                        float d = (z[u] * sc + c_shift[c] > 0.f) ? gg : 0.f;
                        float zh = (z[u] - c_mean[c]) * c_rstd[c];
                        D[r * ldd + c] = sc * (d - c_m1[c] - zh * c_m2[c]);
                    }
                }
            };
            const int step = 256 * U;
            ldg(ga, za, 0);
            for (int base = 0; base < nel; base += 2 * step) {
                if (base + step < nel) ldg(gb, zb, base + step);
                sts(ga, za, base);
                if (base + 2 * step < nel) ldg(ga, za, base + 2 * step);
                if (base + step < nel) sts(gb, zb, base + step);
            }
            const int padc = C4 - C;
            if (padc > 0)
                for (int i = tid; i < nrows * padc; i += 256) {
                    int r = i / padc;
                    D[r * ldd + C + (i - r * padc)] = 0.f;
                }
            for (int i = nrows * C4 + tid; i < 32 * C4; i += 256) {
                int r = i / C4;
                D[r * ldd + (i - r * C4)] = 0.f;
            }
        }
        // ---- stage Zp = raw previous activation (BatchNorm+ReLU applied on the fly in GEMM2) ----
        gg_stage_rows4<false>(Zp, lda, p.Aprev + r0 * cin, nrows, cin, ntn1 * 32, nullptr, nullptr,
                             tid, 256);
        __syncthreads();

        // ---- GEMM1: dX = D * Wb, previous layer's BN-backward sums in the epilogue ----
        if (p.dX) {
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int nt = (t1 >> (8 * j)) & 0xff;
                if (nt == 0xff) continue;
                ggm_f32x16 acc[1];
                ggm_zero<1>(acc);
                if (WLDS) ggm_mma_lds<1>(D, ldd, Wl + nt * C4 * 32, C4, acc);
                else ggm_mma<1>(D, ldd, p.Wb + (size_t)nt * C4 * 32, C4, acc);
                const int col = nt * 32 + (lane & 31);
                if (col < cin) {
                    const float ps = c_ps[col], psh = c_psh[col], pm = c_pm[col], pr = c_pr[col];
                    float a1 = 0.f, a2 = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int row = ggm_row(r, lane);
                        if (row < nrows) {
                            const float dx = acc[0][r];
                            p.dX[(r0 + row) * cin + col] = dx;
                            if (prevbn) {
                                const float zp = Zp[row * lda + col];
                                const float d = (zp * ps + psh > 0.f) ? dx : 0.f;
                                a1 += d;
                                a2 += d * ((zp - pm) * pr);
                            }
                        }
                    }
                    s1[j] += a1;
                    s2[j] += a2;
                }
            }
        }
        // ---- GEMM2: dW(m,n) += act(Zp)^T * D over the 32 rows of the tile ----
#pragma unroll
        for (int j = 0; j < PAIRS; j++) {
            const int q = (t2[j >> 2] >> (8 * (j & 3))) & 0xff;
            if (q == 0xff) continue;
            const int mt = q / ntn2, nt = q - mt * ntn2;
            const int mi = mt * 32 + (lane & 31);            // this lane's cin column
            const bool mok = mi < cin;
            const float ps = mok ? c_ps[mi] : 0.f, psh = mok ? c_psh[mi] : 0.f;
            const float *ap = Zp + (lane >> 5) * lda + mi;
            const float *bp = D + (lane >> 5) * ldd + nt * 32 + (lane & 31);
            const bool nok = nt * 32 + (lane & 31) < C4;
            float av[16], bv[16];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                av[k] = ap[2 * k * lda];
                bv[k] = nok ? bp[2 * k * ldd] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 16; k++) {
                float a = av[k];
                if (prevbn) { a = a * ps + psh; a = a > 0.f ? a : 0.f; }
                if (!mok) a = 0.f;
                accW[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[k], accW[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // ---- flush: dW partials of this workgroup, BN-backward sums of the previous layer ----
    const int cinP = ntn1 * 32, CP = ntn2 * 32;
    float *wpart = p.dWpart + (size_t)blockIdx.x * cinP * CP;
#pragma unroll
    for (int j = 0; j < PAIRS; j++) {
        const int q = (t2[j >> 2] >> (8 * (j & 3))) & 0xff;
        if (q == 0xff) continue;
        const int mt = q / ntn2, nt = q - mt * ntn2;
#pragma unroll
        for (int r = 0; r < 16; r++)
            wpart[(size_t)(mt * 32 + ggm_row(r, lane)) * CP + nt * 32 + (lane & 31)] = accW[j][r];
    }
    if (p.dX && prevbn) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int nt = (t1 >> (8 * j)) & 0xff;
            float a1 = s1[j] + __shfl_xor(s1[j], 32, 64);
            float a2 = s2[j] + __shfl_xor(s2[j], 32, 64);
            const int col = nt * 32 + lane;
            if (nt != 0xff && lane < 32 && col < cin) {
                atomicAdd(&p.psums[col], (double)a1);
                atomicAdd(&p.psums[cin + col], (double)a2);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// dW += act(Aprev)^T * dZ only (split mode: dX comes from gg_k_linear_dx).  With 32-row tiles the
// MFMA work per wave and tile (80 MFMAs, 2.4 us) is far smaller than the fixed cost of staging the
// tile (two HBM round trips + barriers, ~8 us): measured 7.2 ms for 110 GFLOP.  Here a workgroup
// stages RT = 64/128 rows per barrier pair, so the same round trips feed 2-4x the math.
template <int PAIRS>
__global__ __launch_bounds__(256) void gg_k_linear_dw(GGLinBwd p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = p.C, cin = p.cin, ldd = p.ldd, lda = p.lda, RT = p.rt;
    const int C4 = (C + 3) & ~3;
    const int ntm = (cin + 31) >> 5, ntn2 = (C + 31) >> 5, cinP = ntm * 32, CP = ntn2 * 32;
    float *D = lds;                          // [RT][ldd]
    float *Zp = D + RT * ldd;                // [RT][lda]
    float *cst = Zp + RT * lda;
    float *c_scale = cst, *c_shift = cst + C, *c_mean = cst + 2 * C, *c_rstd = cst + 3 * C;
    float *c_m1 = cst + 4 * C, *c_m2 = cst + 5 * C, *c_ps = cst + 6 * C, *c_psh = c_ps + cin;
    const bool prevbn = p.pscale != nullptr;
    for (int c = tid; c < C; c += 256) {
        c_scale[c] = p.scale[c]; c_shift[c] = p.shift[c]; c_mean[c] = p.mean[c];
        c_rstd[c] = p.rstd[c]; c_m1[c] = p.m1[c]; c_m2[c] = p.m2[c];
    }
    for (int c = tid; c < cin; c += 256) {
        c_ps[c] = prevbn ? p.pscale[c] : 1.f; c_psh[c] = prevbn ? p.pshift[c] : 0.f;
    }
    unsigned t2[3];
#pragma unroll
    for (int w = 0; w < 3; w++)
        t2[w] = wave == 0 ? p.t2[0][w] : (wave == 1 ? p.t2[1][w] : (wave == 2 ? p.t2[2][w] : p.t2[3][w]));
    ggm_f32x16 accW[PAIRS];
    ggm_zero<PAIRS>(accW);
    const long long ntile = (p.E + RT - 1) / RT;
    const float invC = 1.0f / (float)C, invP = 1.0f / (float)p.P;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int qsh = __ffs(C) - 3;                    // log2(C/4); C is a power of two >= 4
    const int cc0 = (tid * 4) & (C - 1);
    float k_sc[4], k_sh[4], k_mu[4], k_rs[4], k_m1[4], k_m2[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        k_sc[k] = c_scale[cc0 + k]; k_sh[k] = c_shift[cc0 + k]; k_mu[k] = c_mean[cc0 + k];
        k_rs[k] = c_rstd[cc0 + k]; k_m1[k] = c_m1[cc0 + k]; k_m2[k] = c_m2[cc0 + k];
    }

    for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const long long r0 = tile * RT;
        const int nrows = (p.E - r0 < RT) ? (int)(p.E - r0) : RT;
        // ---- stage D = dZ (16-byte loads, two chunks in flight per thread) ----
        {
            const int nq = (nrows * C) >> 2;
            const float4 *z4 = (const float4 *)(p.Z + r0 * C);
            const float4 *g4 = (const float4 *)(p.dY + r0 * C);
            const long long o0 = p.amax ? r0 / p.P : 0;
            const int rem0 = p.amax ? (int)(r0 - o0 * p.P) : 0;
            constexpr int U = 4;
            float4 za[U], ga[U], zb[U], gb[U];
            auto ldg = [&](float4 (&z)[U], float4 (&g)[U], int base) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int q = base + u * 256 + tid;
                    z[u] = q < nq ? z4[q] : zero4;
                    g[u] = zero4;
                    if (q < nq) {
                        if (p.amax) {
                            const int r = q >> qsh, c = cc0;
                            const int t = rem0 + r;
                            const int oc = (int)(((float)t + 0.5f) * invP);
                            const int pp = t - oc * p.P;
                            const long long idx = (o0 + oc) * C + c;
                            const int4 am = gg_amax4(p.amax + idx);
                            const float4 gv = *(const float4 *)(p.gval + idx);
                            g[u].x = am.x == pp ? gv.x : 0.f; g[u].y = am.y == pp ? gv.y : 0.f;
                            g[u].z = am.z == pp ? gv.z : 0.f; g[u].w = am.w == pp ? gv.w : 0.f;
                        } else {
                            g[u] = g4[q];
                        }
                    }
                }
            };
            auto sts = [&](const float4 (&z)[U], const float4 (&g)[U], int base) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int q = base + u * 256 + tid;
                    if (q < nq) {
                        // 256 threads * 4 floats is a multiple of C (a power of two): this thread
                        // always owns channels cc0..cc0+3, whose constants sit in registers
                        const int r = q >> qsh;
                        const float zv[4] = {z[u].x, z[u].y, z[u].z, z[u].w};
                        const float gv[4] = {g[u].x, g[u].y, g[u].z, g[u].w};
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const float d = (zv[k] * k_sc[k] + k_sh[k] > 0.f) ? gv[k] : 0.f;
                            const float zh = (zv[k] - k_mu[k]) * k_rs[k];
                            D[r * ldd + cc0 + k] = k_sc[k] * (d - k_m1[k] - zh * k_m2[k]);
                        }
                    }
                }
            };
            const int step = 256 * U;
            ldg(za, ga, 0);
            for (int base = 0; base < nq; base += 2 * step) {
                if (base + step < nq) ldg(zb, gb, base + step);
                sts(za, ga, base);
                if (base + 2 * step < nq) ldg(za, ga, base + 2 * step);
                if (base + step < nq) sts(zb, gb, base + step);
            }
            if (nrows < RT)
                for (int i = nrows * C4 + tid; i < RT * C4; i += 256) {
                    int r = i / C4;
                    D[r * ldd + (i - r * C4)] = 0.f;
                }
        }
        // ---- stage Zp = raw previous activation ----
        gg_stage_rows4<false>(Zp, lda, p.Aprev + r0 * cin, nrows, cin, cinP, nullptr, nullptr, tid,
                             256, RT);
        __syncthreads();
        // ---- dW(m,n) += act(Zp)^T * D over the RT rows ----
#pragma unroll
        for (int j = 0; j < PAIRS; j++) {
            const int q = (t2[j >> 2] >> (8 * (j & 3))) & 0xff;
            if (q == 0xff) continue;
            const int mt = q / ntn2, nt = q - mt * ntn2;
            const int mi = mt * 32 + (lane & 31);
            const bool mok = mi < cin;
            const float ps = mok ? c_ps[mi] : 0.f, psh = mok ? c_psh[mi] : 0.f;
            const bool nok = nt * 32 + (lane & 31) < C4;
            for (int rb = 0; rb < nrows; rb += 32) {
                const float *ap = Zp + (rb + (lane >> 5)) * lda + mi;
                const float *bp = D + (rb + (lane >> 5)) * ldd + nt * 32 + (lane & 31);
                float av[16], bv[16];
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    av[k] = ap[2 * k * lda];
                    bv[k] = nok ? bp[2 * k * ldd] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    float a = av[k];
                    if (prevbn) { a = a * ps + psh; a = a > 0.f ? a : 0.f; }
                    if (!mok) a = 0.f;
                    accW[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[k], accW[j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    float *wpart = p.dWpart + (size_t)blockIdx.x * cinP * CP;
#pragma unroll
    for (int j = 0; j < PAIRS; j++) {
        const int q = (t2[j >> 2] >> (8 * (j & 3))) & 0xff;
        if (q == 0xff) continue;
        const int mt = q / ntn2, nt = q - mt * ntn2;
#pragma unroll
        for (int r = 0; r < 16; r++)
            wpart[(size_t)(mt * 32 + ggm_row(r, lane)) * CP + nt * 32 + (lane & 31)] = accW[j][r];
    }
}

// ------------------------------------------------------------------------------------------
// dX = dZ * W of one layer with ONE WAVE PER 32-ROW TILE (persistent, 4 independent waves per
// workgroup, ~17 KB of LDS and < 128 registers each): the monolithic kernel above needs the whole
// register file and > 100 KB of LDS, so only one workgroup fits a CU and every staging round trip
// is exposed (measured: its phases add up serially).  Many light waves let the CU overlap one
// wave's loads with another's MFMAs.  Wg = W packed in column blocks of 4/2/1 tiles.
__global__ __launch_bounds__(256) void gg_k_linear_dx(GGLinBwd p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int C = p.C, cin = p.cin, ldd = p.ldd;
    const int C4 = (C + 3) & ~3;
    const int ntn1 = (cin + 31) >> 5, cinP = ntn1 * 32;
    float *c_scale = lds, *c_shift = lds + C, *c_mean = lds + 2 * C, *c_rstd = lds + 3 * C;
    float *c_m1 = lds + 4 * C, *c_m2 = lds + 5 * C;
    float *Dw = lds + 6 * C4 + wave * (32 * ldd + 2 * cinP);   // [32][ldd] + sums [2][cinP]
    float *sacc = Dw + 32 * ldd;
    for (int c = tid; c < C; c += blockDim.x) {
        c_scale[c] = p.scale[c]; c_shift[c] = p.shift[c]; c_mean[c] = p.mean[c];
        c_rstd[c] = p.rstd[c]; c_m1[c] = p.m1[c]; c_m2[c] = p.m2[c];
    }
    for (int c = lane; c < 2 * cinP; c += 64) sacc[c] = 0.f;
    __syncthreads();
    const bool prevbn = p.pscale != nullptr;
    const long long ntile = (p.E + 31) >> 5;
    const float invC = 1.0f / (float)C, invP = 1.0f / (float)p.P;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int qsh = __ffs(C) - 3;                    // log2(C/4); C is a power of two >= 4
    const int cc0 = (lane * 4) & (C - 1);
    (void)invC;

    for (long long tile = (long long)blockIdx.x * nw + wave; tile < ntile;
         tile += (long long)gridDim.x * nw) {
        const long long r0 = tile << 5;
        const int nrows = (p.E - r0 < 32) ? (int)(p.E - r0) : 32;
        // ---- stage D = dZ of this tile (16-byte loads, two chunks in flight) ----
        {
            const int nq = (nrows * C) >> 2;
            const float4 *z4 = (const float4 *)(p.Z + r0 * C);
            const float4 *g4 = (const float4 *)(p.dY + r0 * C);
            const long long o0 = p.amax ? r0 / p.P : 0;
            const int rem0 = p.amax ? (int)(r0 - o0 * p.P) : 0;
            constexpr int U = 4;
            float4 za[U], ga[U], zb[U], gb[U];
            auto ldg = [&](float4 (&z)[U], float4 (&g)[U], int base) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int q = base + u * 64 + lane;
                    z[u] = q < nq ? z4[q] : zero4;
                    g[u] = zero4;
                    if (q < nq) {
                        if (p.amax) {
                            // sparse upstream gradient: row e = centre e/P, neighbour e%P
                            const int r = q >> qsh, c = cc0;
                            const int t = rem0 + r;
                            const int oc = (int)(((float)t + 0.5f) * invP);
                            const int pp = t - oc * p.P;
                            const long long idx = (o0 + oc) * C + c;
                            const int4 am = gg_amax4(p.amax + idx);
                            const float4 gv = *(const float4 *)(p.gval + idx);
                            g[u].x = am.x == pp ? gv.x : 0.f; g[u].y = am.y == pp ? gv.y : 0.f;
                            g[u].z = am.z == pp ? gv.z : 0.f; g[u].w = am.w == pp ? gv.w : 0.f;
                        } else {
                            g[u] = g4[q];
                        }
                    }
                }
            };
            auto sts = [&](const float4 (&z)[U], const float4 (&g)[U], int base) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int q = base + u * 64 + lane;
                    if (q < nq) {
                        // 64 lanes * 4 floats is a multiple of C (a power of two <= 256): this lane
                        // always owns channels cc0..cc0+3, whose constants sit in registers
                        const int r = q >> qsh;
                        const float zv[4] = {z[u].x, z[u].y, z[u].z, z[u].w};
                        const float gv[4] = {g[u].x, g[u].y, g[u].z, g[u].w};
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            // (constants from LDS: keeping 24 of them in registers cost this
                            //  light kernel occupancy -- measured 2.9 -> 3.7 ms)
                            const int cc = cc0 + k;
                            const float sc = c_scale[cc];
                            const float d = (zv[k] * sc + c_shift[cc] > 0.f) ? gv[k] : 0.f;
                            const float zh = (zv[k] - c_mean[cc]) * c_rstd[cc];
                            Dw[r * ldd + cc] = sc * (d - c_m1[cc] - zh * c_m2[cc]);
                        }
                    }
                }
            };
            const int step = 64 * U;
            ldg(za, ga, 0);
            for (int base = 0; base < nq; base += 2 * step) {
                if (base + step < nq) ldg(zb, gb, base + step);
                sts(za, ga, base);
                if (base + 2 * step < nq) ldg(za, ga, base + 2 * step);
                if (base + step < nq) sts(zb, gb, base + step);
            }
            gg_tile_pad(Dw, ldd, nrows, C, C4, lane, 64);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- dX tile = D * W, column blocks of 4 / 2 / 1 tiles ----
        int done = 0;
        const float *wg = p.Wg;
        while (done < ntn1) {
            const int rem = ntn1 - done;
            const int nt = rem >= 4 ? 4 : (rem >= 2 ? 2 : 1);
            ggm_f32x16 acc[4];
            ggm_zero<4>(acc);
            if (nt == 4) ggm_mma<4>(Dw, ldd, wg, C4, acc);
            else if (nt == 2) {
                ggm_f32x16 a2[2];
                ggm_zero<2>(a2);
                ggm_mma<2>(Dw, ldd, wg, C4, a2);
                acc[0] = a2[0]; acc[1] = a2[1];
            } else {
                ggm_f32x16 a1[1];
                ggm_zero<1>(a1);
                ggm_mma<1>(Dw, ldd, wg, C4, a1);
                acc[0] = a1[0];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (j >= nt) continue;
                const int col = (done + j) * 32 + (lane & 31);
                float a1 = 0.f, a2 = 0.f;
                if (col < cin) {
                    const float ps = prevbn ? p.pscale[col] : 0.f, psh = prevbn ? p.pshift[col] : 0.f;
                    const float pm = prevbn ? p.pmean[col] : 0.f, pr = prevbn ? p.prstd[col] : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int row = ggm_row(r, lane);
                        if (row < nrows) {
                            const float dx = acc[j][r];
                            p.dX[(r0 + row) * cin + col] = dx;
                            if (prevbn) {
                                const float zp = p.Aprev[(r0 + row) * cin + col];
                                const float d = (zp * ps + psh > 0.f) ? dx : 0.f;
                                a1 += d;
                                a2 += d * ((zp - pm) * pr);
                            }
                        }
                    }
                }
                if (prevbn) {
                    a1 += __shfl_xor(a1, 32, 64);
                    a2 += __shfl_xor(a2, 32, 64);
                    if (lane < 32) { sacc[col] += a1; sacc[cinP + col] += a2; }
                }
            }
            wg += (size_t)C4 * 32 * nt;
            done += nt;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (prevbn)
        for (int c = lane; c < cin; c += 64) {
            atomicAdd(&p.psums[c], (double)sacc[c]);
            atomicAdd(&p.psums[cin + c], (double)sacc[cinP + c]);
        }
}

// dW[c][i] = sum_wg part[wg][i][c]   (part: [nwg][cinP][CP]; dW: torch layout [C][cin])
// block = 64 elements x 4 slices of the workgroup range
__global__ __launch_bounds__(256) void gg_k_dw_reduce(const float *__restrict__ part, int nwg,
                                                      int cinP, int CP, int cin, int C, int cin_w,
                                                      int rot, float *__restrict__ dW)
{
    __shared__ float sh[256];
    const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;
    const size_t S = (size_t)cinP * CP;
    float s = 0.f;
    if (e < (int)S)
        for (int w = sl; w < nwg; w += 4) s += part[(size_t)w * S + e];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (sl == 0 && e < (int)S) {
        s = sh[el] + sh[64 + el] + sh[128 + el] + sh[192 + el];
        const int i = e / CP, c = e - i * CP;
        // framework layout [C][cin_w]: zero-padding columns dropped, rotated columns moved back
        if (i < cin && i < cin_w && c < C) {
            const int f = i < cin_w - rot ? i + rot : i - (cin_w - rot);
            dW[(size_t)c * cin_w + f] = s;
        }
    }
}

int gg_linear_bwd_workspace(long long E, int cin, int C, size_t *bytes, int *nwg)
{
    // upper bound: up to 4 persistent workgroups per CU (the launcher picks by LDS footprint)
    long long ntile = (E + 31) >> 5;
    int n = (int)(ntile < 1024 ? ntile : 1024);
    const int cinP = ((cin + 31) >> 5) * 32, CP = ((C + 31) >> 5) * 32;
    if (nwg) *nwg = n;
    if (bytes) {
        *bytes = (size_t)n * cinP * CP * sizeof(float);
        const size_t d = gg_linear_dw_direct_workspace(E, cin, C);
        if (d > *bytes) *bytes = d;
        const size_t f = gg_att_bwd_fused_workspace(E, cin, C);
        if (f > *bytes) *bytes = f;
        const size_t f2 = gg_linear_bwd_fused128_workspace(E);
        if (C == 128 && (cin == 128 || cin == 256) && f2 > *bytes) *bytes = f2;
    }
    return 0;
}

template <int PAIRS>
static int launch_bwd(const GGLinBwd &p, bool wlds, size_t lds, int nwg, hipStream_t st)
{
    static GGDevOnce attr_done;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)gg_k_linear_bwd<PAIRS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 3;
        if (hipFuncSetAttribute((const void *)gg_k_linear_bwd<PAIRS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 3;
        attr_done = true;
    }
    if (wlds) gg_k_linear_bwd<PAIRS, true><<<nwg, 256, lds, st>>>(p);
    else gg_k_linear_bwd<PAIRS, false><<<nwg, 256, lds, st>>>(p);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_linear_bwd(const GGLinBwd &pin, hipStream_t st)
{
    GGLinBwd p = pin;
    if (p.E < 1 || p.C < 1 || p.C > 256 || p.cin < 1 || p.cin > 384) return 1;
    const int C4 = (p.C + 3) & ~3;
    const int ntm = (p.cin + 31) >> 5, ntn2 = (p.C + 31) >> 5;
    p.ldd = C4 | 1;
    p.lda = (ntm * 32) | 1;
    // ---- 32 -> 64/128 layer behind a BatchNorm'd layer: dX, sums and dW in one pass over Z ----
    // (a bf16 Z has no other reader: it takes the fused kernel whatever the option says NOW -- the
    //  storage format was decided at forward time, when the option was on)
    if (p.dX && p.Wdx && (g_opt_att_bwd_fused || p.zfmt)) {
        const int rc = gg_att_bwd_fused(p, st);
        if (rc != 1) return rc;
    }
    if (p.zfmt) return 1;   // a bf16 Z is only read by the fused attention backward
    // ---- 128-output per-point layer, dense gradient: dX, dW and the sums from ONE pass over Z and dY ----
    {
        const int rc = gg_linear_bwd_fused128(p, st);
        if (rc != 1) return rc;
    }
    // ---- register-direct dX (gridgcn_direct.hip) when the operand was packed for it ----
    if (p.dX && p.Wdx) {
        const int rc = gg_linear_dx_direct(p, st);
        if (rc == 0) p.dX = nullptr;
        else if (rc != 1) return rc;
    }
    // a strided dense dY / a strided Z is only understood by the register-direct kernels
    if (p.dX && !p.amax && p.ldy != p.C) return 1;
    // from here on a kernel that reads the m1 / m2 ARRAYS may run: GGLinBwd.bsums -> the vectors first
    auto finalize_now = [&]() -> int {
        if (!p.bsums) return 0;
        const int rc = gg_bn_bwd_finalize(p.bsums, p.E, p.C, p.fin_m1, p.fin_m2, p.fin_dgamma, p.fin_dbeta, st);
        p.m1 = p.fin_m1; p.m2 = p.fin_m2;
        p.bsums = nullptr;
        return rc;
    };
    if (p.dX && p.ldz && p.ldz != p.C) return 1;
    // ---- split mode: dX by the light one-wave-per-tile kernel, then dW by the kernel below ----
    if (p.dX && p.Wg && p.C >= 4 && (p.C & (p.C - 1)) == 0) {
        if (finalize_now()) return 3;
        static GGDevOnce attr_dx;
        if (!attr_dx) {
            if (hipFuncSetAttribute((const void *)gg_k_linear_dx, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 3;
            attr_dx = true;
        }
        const int nw = 4;
        const size_t ldsx = ((size_t)6 * C4 + (size_t)nw * (32 * p.ldd + 2 * ntm * 32)) * sizeof(float);
        if (ldsx <= 150 * 1024) {
            int per_cu = (int)((150 * 1024) / ldsx);
            per_cu = per_cu < 1 ? 1 : (per_cu > 3 ? 3 : per_cu);
            const long long ntile = (p.E + 31) >> 5;
            long long nb = (ntile + nw - 1) / nw;
            if (nb > 256 * per_cu) nb = 256 * per_cu;
            gg_k_linear_dx<<<(int)nb, 64 * nw, ldsx, st>>>(p);
            if (hipGetLastError() != hipSuccess) return 3;
            p.dX = nullptr;                      // the kernel below only accumulates dW now
        }
    }
    // ---- register-direct dW (gridgcn_direct.hip) once dX is out of the way ----
    if (!p.dX) {
        const int rc = gg_linear_dw_direct(p, st);
        if (rc != 1) return rc;
    }
    if (!p.amax && p.ldy != p.C) return 1;
    if (p.ldz && p.ldz != p.C) return 1;
    if (finalize_now()) return 3;
    const int npairs = ntm * ntn2;
    if (npairs > 48) return 1;
    // ---- balance GEMM1 column tiles (cost C4/2 MFMAs) and GEMM2 pairs (16 MFMAs) over 4 waves ----
    int load[4] = {0, 0, 0, 0}, n1[4] = {0, 0, 0, 0}, n2[4] = {0, 0, 0, 0};
    unsigned char l1[4][3], l2[4][12];
    for (int w = 0; w < 4; w++) { for (int j = 0; j < 3; j++) l1[w][j] = 0xff; for (int j = 0; j < 12; j++) l2[w][j] = 0xff; }
    if (p.dX)
        for (int t = 0; t < ntm; t++) {
            int w = 0;
            for (int x = 1; x < 4; x++) if (load[x] < load[w]) w = x;
            if (n1[w] >= 3) return 1;
            l1[w][n1[w]++] = (unsigned char)t;
            load[w] += C4 / 2;
        }
    for (int q = 0; q < npairs; q++) {
        int w = -1;
        for (int x = 0; x < 4; x++) if (n2[x] < 12 && (w < 0 || load[x] < load[w])) w = x;
        if (w < 0) return 1;
        l2[w][n2[w]++] = (unsigned char)q;
        load[w] += 16;
    }
    int pmax = 0;
    for (int w = 0; w < 4; w++) {
        if (n2[w] > pmax) pmax = n2[w];
        p.t1[w] = l1[w][0] | (l1[w][1] << 8) | (l1[w][2] << 16) | 0xff000000u;
        for (int g = 0; g < 3; g++)
            p.t2[w][g] = l2[w][4 * g] | (l2[w][4 * g + 1] << 8) | (l2[w][4 * g + 2] << 16) |
                         ((unsigned)l2[w][4 * g + 3] << 24);
    }
    p.ncen_max = p.amax ? (31 + p.P - 1) / p.P + 2 : 0;   // centres a 32-row tile can touch
    const size_t base = ((size_t)32 * p.ldd + (size_t)32 * p.lda + 6 * p.C + 4 * p.cin +
                         2 * (size_t)p.ncen_max * p.C) * sizeof(float);
    const size_t wbytes = (size_t)ntm * C4 * 32 * sizeof(float);
    // weights resident in LDS only when that still leaves room for >= 2 workgroups per CU: with a
    // single workgroup per CU nothing overlaps the staging round trips of a tile
    const size_t wcap = 158 * 1024;
    const bool wlds = p.dX && (base + wbytes <= wcap);
    const size_t lds = base + (wlds ? wbytes : 0);
    if (lds > 158 * 1024) return 1;
    int nwg;
    gg_linear_bwd_workspace(p.E, p.cin, p.C, nullptr, &nwg);
    {
        int per_cu = (int)((150 * 1024) / lds);
        per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
        if (nwg > 256 * per_cu) nwg = 256 * per_cu;
    }
    int rc;
    // ---- dW-only kernel with large row tiles (split mode, or no input gradient needed) ----
    if (!p.dX && p.C >= 4 && (p.C & (p.C - 1)) == 0) {
        const size_t cbytes = ((size_t)6 * p.C + 2 * p.cin) * sizeof(float);
        int rt = 0;
        const int cands[4] = {128, 96, 64, 32};
        for (int k = 0; k < 4 && !rt; k++)
            if ((size_t)cands[k] * (p.ldd + p.lda) * 4 + cbytes <= 150 * 1024) rt = cands[k];
        if (rt) {
            p.rt = rt;
            const size_t ldsw = (size_t)rt * (p.ldd + p.lda) * 4 + cbytes;
            const long long ntile = (p.E + rt - 1) / rt;
            int per_cu = (int)((150 * 1024) / ldsw);
            per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
            int nw2 = (int)(ntile < 256 * per_cu ? ntile : 256 * per_cu);
            if (nw2 > nwg) nw2 = nwg;             // workspace was sized for nwg partials
#define GG_DW(PP)                                                                               \
    do {                                                                                        \
        static GGDevOnce done_##PP;                                                                  \
        if (!done_##PP) {                                                                       \
            if (hipFuncSetAttribute((const void *)gg_k_linear_dw<PP>,                           \
                                    hipFuncAttributeMaxDynamicSharedMemorySize,                 \
                                    160 * 1024) != hipSuccess) return 3;                        \
            done_##PP = true;                                                                   \
        }                                                                                       \
        gg_k_linear_dw<PP><<<nw2, 256, ldsw, st>>>(p);                                          \
    } while (0)
            if (pmax <= 2) GG_DW(2);
            else if (pmax <= 5) GG_DW(5);
            else if (pmax <= 8) GG_DW(8);
            else GG_DW(12);
#undef GG_DW
            if (hipGetLastError() != hipSuccess) return 3;
            const int cinP2 = ntm * 32, CP2 = ntn2 * 32;
            gg_k_dw_reduce<<<(cinP2 * CP2 + 63) / 64, 256, 0, st>>>(p.dWpart, nw2, cinP2, CP2, p.cin,
                                                                    p.C, p.cin_w, p.rot, p.dW);
            return hipGetLastError() == hipSuccess ? 0 : 3;
        }
    }
    if (pmax <= 1) rc = launch_bwd<1>(p, wlds, lds, nwg, st);
    else if (pmax <= 2) rc = launch_bwd<2>(p, wlds, lds, nwg, st);
    else if (pmax <= 3) rc = launch_bwd<3>(p, wlds, lds, nwg, st);
    else if (pmax <= 5) rc = launch_bwd<5>(p, wlds, lds, nwg, st);
    else if (pmax <= 7) rc = launch_bwd<7>(p, wlds, lds, nwg, st);
    else if (pmax <= 9) rc = launch_bwd<9>(p, wlds, lds, nwg, st);
    else rc = launch_bwd<12>(p, wlds, lds, nwg, st);
    if (rc) return rc;
    const int cinP = ntm * 32, CP = ntn2 * 32;
    gg_k_dw_reduce<<<(cinP * CP + 63) / 64, 256, 0, st>>>(p.dWpart, nwg, cinP, CP, p.cin, p.C,
                                                          p.cin_w, p.rot, p.dW);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

// ------------------------------------------------------------------------------------------
// Y = relu(Z*scale[c] + shift[c]);  C % 4 == 0 -> float4
__global__ __launch_bounds__(256) void gg_k_bn_apply(const float *__restrict__ Z,
                                                     const float *__restrict__ scale,
                                                     const float *__restrict__ shift,
                                                     float *__restrict__ Y, long long total, int C,
                                                     int ldy, unsigned thr, float dscale,
                                                     unsigned slo, unsigned shi,
                                                     const unsigned long long *__restrict__ sdev)
{
    if (sdev) {  // graph replay: the seed advances through a device scalar
        const unsigned long long sd = (((unsigned long long)shi << 32) | slo) + *sdev;
        slo = (unsigned)sd;
        shi = (unsigned)(sd >> 32);
    }
    // thr != 0: Dropout behind the ReLU (mx.sym.Dropout, ggcn_models_g.py:36): kept values * dscale
    if ((C & 3) == 0 && (ldy & 3) == 0) {
        const long long n4 = total >> 2;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4;
             i += (long long)gridDim.x * 256) {
            float4 z = ((const float4 *)Z)[i];
            const long long row = (i * 4) / C;
            const int c = (int)(i * 4 - row * C);
            const float4 sc = *(const float4 *)(scale + c), sh = *(const float4 *)(shift + c);
            float4 y;
            y.x = fmaxf(z.x * sc.x + sh.x, 0.f); y.y = fmaxf(z.y * sc.y + sh.y, 0.f);
            y.z = fmaxf(z.z * sc.z + sh.z, 0.f); y.w = fmaxf(z.w * sc.w + sh.w, 0.f);
            if (thr) {
                const unsigned long long e = (unsigned long long)i * 4;
                y.x = gg_drop_keep(e, slo, shi, thr) ? y.x * dscale : 0.f;
                y.y = gg_drop_keep(e + 1, slo, shi, thr) ? y.y * dscale : 0.f;
                y.z = gg_drop_keep(e + 2, slo, shi, thr) ? y.z * dscale : 0.f;
                y.w = gg_drop_keep(e + 3, slo, shi, thr) ? y.w * dscale : 0.f;
            }
            *(float4 *)(Y + row * ldy + c) = y;
        }
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
             i += (long long)gridDim.x * 256) {
            const long long row = i / C;
            const int c = (int)(i - row * C);
            float y = fmaxf(Z[i] * scale[c] + shift[c], 0.f);
            if (thr) y = gg_drop_keep((unsigned long long)i, slo, shi, thr) ? y * dscale : 0.f;
            Y[row * ldy + c] = y;
        }
    }
}

// s1[c] = sum dyr, s2[c] = sum dyr * zhat, dyr = dY * (Z*scale+shift > 0), zhat = (Z-mean)*rstd
// thread = (row phase tid / C, channel tid % C); requires 256 % C == 0 or C % 256 == 0.
__global__ __launch_bounds__(256) void gg_k_bn_bwd_reduce(const float *__restrict__ dY,
                                                          const float *__restrict__ Z,
                                                          const float *__restrict__ scale,
                                                          const float *__restrict__ shift,
                                                          const float *__restrict__ mean,
                                                          const float *__restrict__ rstd,
                                                          long long E, int C,
                                                          double *__restrict__ sums, int ldy)
{
    __shared__ float sh1[256], sh2[256];
    const int tid = threadIdx.x;
    if (C <= 256) {
        const int rpp = 256 / C;                      // rows per pass
        const int c = tid % C, rr = tid / C;
        const float sc = scale[c], sf = shift[c], mu = mean[c], rs = rstd[c];
        float a1 = 0.f, a2 = 0.f;
        for (long long r = (long long)blockIdx.x * rpp + rr; r < E; r += (long long)gridDim.x * rpp) {
            const float z = Z[r * C + c];
            const float d = (z * sc + sf > 0.f) ? dY[r * ldy + c] : 0.f;
            a1 += d;
            a2 += d * ((z - mu) * rs);
        }
        sh1[tid] = a1; sh2[tid] = a2;
        __syncthreads();
        if (tid < C) {
            for (int j = 1; j < rpp; j++) { a1 += sh1[tid + j * C]; a2 += sh2[tid + j * C]; }
            atomicAdd(&sums[c], (double)a1);
            atomicAdd(&sums[C + c], (double)a2);
        }
    } else {
        for (int c = tid; c < C; c += 256) {
            const float sc = scale[c], sf = shift[c], mu = mean[c], rs = rstd[c];
            float a1 = 0.f, a2 = 0.f;
            for (long long r = blockIdx.x; r < E; r += gridDim.x) {
                const float z = Z[r * C + c];
                const float d = (z * sc + sf > 0.f) ? dY[r * ldy + c] : 0.f;
                a1 += d;
                a2 += d * ((z - mu) * rs);
            }
            atomicAdd(&sums[c], (double)a1);
            atomicAdd(&sums[C + c], (double)a2);
        }
    }
}

// dZ = a[c] * (dyr - m1[c] - zhat*m2[c])
__global__ __launch_bounds__(256) void gg_k_bn_bwd_elemt(const float *__restrict__ dY,
                                                         const float *__restrict__ Z,
                                                         const float *__restrict__ scale,
                                                         const float *__restrict__ shift,
                                                         const float *__restrict__ mean,
                                                         const float *__restrict__ rstd,
                                                         const float *__restrict__ m1,
                                                         const float *__restrict__ m2,
                                                         long long total, int C,
                                                         float *__restrict__ dZ)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
         i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const float z = Z[i];
        const float sc = scale[c];
        const float d = (z * sc + shift[c] > 0.f) ? dY[i] : 0.f;
        const float zh = (z - mean[c]) * rstd[c];
        dZ[i] = sc * (d - m1[c] - zh * m2[c]);      // scale = gamma * rstd
    }
}

static int grid_for(long long work, int per_block, int cap)
{
    long long nb = (work + per_block - 1) / per_block;
    return (int)(nb < 1 ? 1 : (nb > cap ? cap : nb));
}

// drop probability p -> (threshold on the 32-bit hash, scale of the kept values)
void gg_drop_consts(float p, unsigned *thr, float *dscale)
{
    double t = (double)p * 4294967296.0;
    *thr = p > 0.f ? (unsigned)(t > 4294967295.0 ? 4294967295.0 : t) : 0u;
    *dscale = p > 0.f ? (float)(1.0 / (1.0 - (double)p)) : 1.f;
}

int gg_bn_apply(const float *Z, const float *scale, const float *shift, float *Y, long long E,
                int C, int ldy, float drop_p, unsigned long long seed,
                const unsigned long long *seed_dev, hipStream_t st)
{
    long long total = E * C;
    unsigned thr;
    float ds;
    gg_drop_consts(drop_p, &thr, &ds);
    gg_k_bn_apply<<<grid_for(total / 4 + 1, 256, 65536), 256, 0, st>>>(
        Z, scale, shift, Y, total, C, ldy, thr, ds, (unsigned)seed, (unsigned)(seed >> 32), seed_dev);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_bn_bwd_reduce(const float *dY, const float *Z, const float *scale, const float *shift,
                     const float *mean, const float *rstd, long long E, int C, double *sums,
                     int ldy, hipStream_t st)
{
    if (!((C <= 256 && 256 % C == 0) || (C % 256 == 0))) return 1;
    int rpp = C <= 256 ? 256 / C : 1;
    gg_k_bn_bwd_reduce<<<grid_for(E, rpp * 16, 2048), 256, 0, st>>>(dY, Z, scale, shift, mean, rstd,
                                                                     E, C, sums, ldy);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_bn_bwd_elemt(const float *dY, const float *Z, const float *scale, const float *shift,
                    const float *mean, const float *rstd, const float *m1, const float *m2,
                    long long E, int C, float *dZ, hipStream_t st)
{
    long long total = E * C;
    gg_k_bn_bwd_elemt<<<grid_for(total, 256, 65536), 256, 0, st>>>(dY, Z, scale, shift, mean, rstd,
                                                                   m1, m2, total, C, dZ);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
