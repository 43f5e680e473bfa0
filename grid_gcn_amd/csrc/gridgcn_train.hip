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
#include "gridgcn_mma.h"
#include "gridgcn_train.h"

// ------------------------------------------------------------------------------------------
// copy a [nrows x cin] row-major chunk (contiguous in global memory) into an LDS tile with row
// stride ld, optionally through x -> relu(x*scale[c] + shift[c]); zero the K padding / missing rows.
template <bool XFORM>
__device__ __forceinline__ void gg_stage_rows(float *dst, int ld, const float *__restrict__ src,
                                              int nrows, int cin, int K,
                                              const float *__restrict__ scale,
                                              const float *__restrict__ shift, int tid, int nthr)
{
    const int nel = nrows * cin;
    const float inv = 1.0f / (float)cin;
    for (int base = 0; base < nel; base += nthr * 4) {
        float v[4];
        int idx[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            idx[u] = base + u * nthr + tid;
            v[u] = idx[u] < nel ? src[idx[u]] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (idx[u] < nel) {
                int r = (int)(((float)idx[u] + 0.5f) * inv);     // exact for nel <= 32768
                int c = idx[u] - r * cin;
                float x = v[u];
                if (XFORM) { x = x * scale[c] + shift[c]; x = x > 0.f ? x : 0.f; }
                dst[r * ld + c] = x;
            }
        }
    }
    const int padc = K - cin;
    if (padc > 0)
        for (int i = tid; i < nrows * padc; i += nthr) {
            int r = i / padc, c = cin + (i - r * padc);
            dst[r * ld + c] = 0.f;
        }
    for (int i = nrows * K + tid; i < 32 * K; i += nthr) {
        int r = i / K, c = i - r * K;
        dst[r * ld + c] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// forward: persistent single-wave workgroups; wave w handles row tiles w, w+gridDim.x, ...
template <int NT>
__global__ __launch_bounds__(64) void gg_k_linear_fwd(GGLinFwd p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [32][lda]
    const int lane = threadIdx.x;
    const int lda = p.lda;
    const long long ntile = (p.E + 31) >> 5;
    const int ngroup = p.ldw / (32 * NT);
    float ssum[2][NT], ssq[2][NT];
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) { ssum[g][nt] = 0.f; ssq[g][nt] = 0.f; }

    for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const long long r0 = tile << 5;
        const int nrows = (p.E - r0 < 32) ? (int)(p.E - r0) : 32;
        if (p.scale) gg_stage_rows<true>(lds, lda, p.X + r0 * p.cin, nrows, p.cin, p.K, p.scale, p.shift, lane, 64);
        else gg_stage_rows<false>(lds, lda, p.X + r0 * p.cin, nrows, p.cin, p.K, nullptr, nullptr, lane, 64);
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 2; g++) {
            if (g >= ngroup) continue;
            ggm_f32x16 acc[NT];
            ggm_zero<NT>(acc);
            ggm_mma<NT>(lds, lda, p.W + (size_t)p.K * g * 32 * NT, p.K, acc);
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
                        p.Z[(r0 + row) * p.cout + col] = z;
                        s += z;
                        q += z * z;
                    }
                }
                ssum[g][nt] += s;
                ssq[g][nt] += q;
            }
        }
        __syncthreads();
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

int gg_linear_fwd(const GGLinFwd &p, hipStream_t st)
{
    if (p.E < 1 || p.cin < 1 || p.K < 4 || (p.K & 3) || p.K < p.cin || p.cin > 1024) return 1;
    if (p.ldw != 32 && p.ldw != 64 && p.ldw != 128 && p.ldw != 256) return 1;
    GGLinFwd q = p;
    q.lda = p.K | 1;
    size_t lds = (size_t)32 * q.lda * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute((const void *)gg_k_linear_fwd<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
        hipFuncSetAttribute((const void *)gg_k_linear_fwd<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
        hipFuncSetAttribute((const void *)gg_k_linear_fwd<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
        attr_done = true;
    }
    long long ntile = (p.E + 31) >> 5;
    int grid = (int)(ntile < 4096 ? ntile : 4096);
    if (p.ldw == 32) gg_k_linear_fwd<1><<<grid, 64, lds, st>>>(q);
    else if (p.ldw == 64) gg_k_linear_fwd<2><<<grid, 64, lds, st>>>(q);
    else gg_k_linear_fwd<4><<<grid, 64, lds, st>>>(q);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

// ------------------------------------------------------------------------------------------
// backward of one (linear -> BatchNorm(batch stats) -> ReLU) layer.  256 threads = 4 waves share
// one 32-row tile; persistent workgroups (tile = blockIdx.x, += gridDim.x).
//   D  [32][ldd] = dZ tile        Zp [32][lda] = raw previous activation tile (Z_{l-1} or X)
//   GEMM1: dX[32 x cin] = D * Wb            column tiles wave, wave+4, ...   (<= 3 per wave)
//   GEMM2: dW[cin x C] += act(Zp)^T * D     (m,n) tile pairs wave, wave+4, ... (<= PAIRS per wave)
template <int PAIRS>
__global__ __launch_bounds__(256) void gg_k_linear_bwd(GGLinBwd p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = p.C, cin = p.cin;
    const int ldd = p.ldd, lda = p.lda;
    float *D = lds;                          // [32][ldd]
    float *Zp = D + 32 * ldd;                // [32][lda]
    float *cst = Zp + 32 * lda;              // per-channel constants
    float *c_scale = cst, *c_shift = cst + C, *c_mean = cst + 2 * C, *c_rstd = cst + 3 * C;
    float *c_m1 = cst + 4 * C, *c_m2 = cst + 5 * C;
    float *c_ps = cst + 6 * C, *c_psh = c_ps + cin, *c_pm = c_psh + cin, *c_pr = c_pm + cin;
    for (int c = tid; c < C; c += 256) {
        c_scale[c] = p.scale[c]; c_shift[c] = p.shift[c]; c_mean[c] = p.mean[c];
        c_rstd[c] = p.rstd[c]; c_m1[c] = p.m1[c]; c_m2[c] = p.m2[c];
    }
    const bool prevbn = p.pscale != nullptr;
    for (int c = tid; c < cin; c += 256) {
        c_ps[c] = prevbn ? p.pscale[c] : 1.f; c_psh[c] = prevbn ? p.pshift[c] : 0.f;
        c_pm[c] = prevbn ? p.pmean[c] : 0.f; c_pr[c] = prevbn ? p.prstd[c] : 0.f;
    }
    const int C4 = (C + 3) & ~3;
    const int ntn1 = (cin + 31) >> 5;        // GEMM1 column tiles (over cin)
    const int ntm = ntn1;                    // GEMM2 M tiles (over cin)
    const int ntn2 = (C + 31) >> 5;          // GEMM2 N tiles (over C)
    const int npairs = ntm * ntn2;
    const long long ntile = (p.E + 31) >> 5;

    ggm_f32x16 accW[PAIRS];
    ggm_zero<PAIRS>(accW);
    float s1[3] = {0.f, 0.f, 0.f}, s2[3] = {0.f, 0.f, 0.f};
    __syncthreads();

    for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const long long r0 = tile << 5;
        const int nrows = (p.E - r0 < 32) ? (int)(p.E - r0) : 32;
        // ---- stage D = dZ (BatchNorm+ReLU backward, element-wise part) ----
        {
            const float *dy = p.dY + r0 * C, *zz = p.Z + r0 * C;
            const int nel = nrows * C;
            const float inv = 1.0f / (float)C;
            for (int base = 0; base < nel; base += 1024) {
                float g[4], z[4];
                int idx[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    idx[u] = base + u * 256 + tid;
                    g[u] = idx[u] < nel ? dy[idx[u]] : 0.f;
                    z[u] = idx[u] < nel ? zz[idx[u]] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (idx[u] < nel) {
                        int r = (int)(((float)idx[u] + 0.5f) * inv);
                        int c = idx[u] - r * C;
                        float sc = c_scale[c];
                        float d = (z[u] * sc + c_shift[c] > 0.f) ? g[u] : 0.f;
                        float zh = (z[u] - c_mean[c]) * c_rstd[c];
                        D[r * ldd + c] = sc * (d - c_m1[c] - zh * c_m2[c]);
                    }
                }
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
        gg_stage_rows<false>(Zp, lda, p.Aprev + r0 * cin, nrows, cin, ntm * 32 < lda ? ntm * 32 : lda - 1,
                             nullptr, nullptr, tid, 256);
        __syncthreads();

        // ---- GEMM1: dX = D * Wb, previous layer's BN-backward sums in the epilogue ----
        if (p.dX) {
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int nt = wave + 4 * j;
                if (nt >= ntn1) continue;
                ggm_f32x16 acc[1];
                ggm_zero<1>(acc);
                ggm_mma<1>(D, ldd, p.Wb + (size_t)nt * C4 * 32, C4, acc);
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
            const int q = wave + 4 * j;
            if (q >= npairs) continue;
            const int mt = q / ntn2, nt = q - mt * ntn2;
            const int mi = mt * 32 + (lane & 31);            // this lane's cin column
            const float ps = mi < cin ? c_ps[mi] : 0.f, psh = mi < cin ? c_psh[mi] : 0.f;
            const float *ap = Zp + (lane >> 5) * lda + (mi < lda ? mi : 0);
            const float *bp = D + (lane >> 5) * ldd + nt * 32 + (lane & 31);
#pragma unroll 4
            for (int k = 0; k < 32; k += 2) {
                float a = ap[k * lda];
                if (prevbn) { a = a * ps + psh; a = a > 0.f ? a : 0.f; }
                if (mi >= cin) a = 0.f;
                const float b = (nt * 32 + (lane & 31) < C4) ? bp[k * ldd] : 0.f;
                accW[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, accW[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // ---- flush: dW partials of this workgroup, BN-backward sums of the previous layer ----
    const int cinP = ntm * 32, CP = ntn2 * 32;
    float *wpart = p.dWpart + (size_t)blockIdx.x * cinP * CP;
#pragma unroll
    for (int j = 0; j < PAIRS; j++) {
        const int q = wave + 4 * j;
        if (q >= npairs) continue;
        const int mt = q / ntn2, nt = q - mt * ntn2;
#pragma unroll
        for (int r = 0; r < 16; r++)
            wpart[(size_t)(mt * 32 + ggm_row(r, lane)) * CP + nt * 32 + (lane & 31)] = accW[j][r];
    }
    if (p.dX && prevbn) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int nt = wave + 4 * j;
            float a1 = s1[j] + __shfl_xor(s1[j], 32, 64);
            float a2 = s2[j] + __shfl_xor(s2[j], 32, 64);
            const int col = nt * 32 + lane;
            if (nt < ntn1 && lane < 32 && col < cin) {
                atomicAdd(&p.psums[col], (double)a1);
                atomicAdd(&p.psums[cin + col], (double)a2);
            }
        }
    }
}

// dW[c][i] = sum_wg part[wg][i][c]   (part: [nwg][cinP][CP]; dW: torch layout [C][cin])
__global__ __launch_bounds__(256) void gg_k_dw_reduce(const float *__restrict__ part, int nwg,
                                                      int cinP, int CP, int cin, int C,
                                                      float *__restrict__ dW)
{
    const int e = blockIdx.x * 256 + threadIdx.x;           // over cinP*CP
    if (e >= cinP * CP) return;
    const int i = e / CP, c = e - i * CP;
    if (i >= cin || c >= C) return;
    float s = 0.f;
    for (int w = 0; w < nwg; w++) s += part[(size_t)w * cinP * CP + e];
    dW[(size_t)c * cin + i] = s;
}

int gg_linear_bwd_workspace(long long E, int cin, int C, size_t *bytes, int *nwg)
{
    long long ntile = (E + 31) >> 5;
    int n = (int)(ntile < 512 ? ntile : 512);
    const int cinP = ((cin + 31) >> 5) * 32, CP = ((C + 31) >> 5) * 32;
    if (nwg) *nwg = n;
    if (bytes) *bytes = (size_t)n * cinP * CP * sizeof(float);
    return 0;
}

int gg_linear_bwd(const GGLinBwd &pin, hipStream_t st)
{
    GGLinBwd p = pin;
    if (p.E < 1 || p.C < 1 || p.C > 256 || p.cin < 1 || p.cin > 1024) return 1;
    const int C4 = (p.C + 3) & ~3;
    const int ntm = (p.cin + 31) >> 5, ntn2 = (p.C + 31) >> 5;
    if (ntm > 12) return 1;                          // 3 column tiles per wave
    p.ldd = C4 | 1;
    p.lda = (ntm * 32) | 1;
    const int npairs = ntm * ntn2;
    const int pairs_per_wave = (npairs + 3) / 4;
    size_t lds = ((size_t)32 * p.ldd + (size_t)32 * p.lda + 6 * p.C + 4 * p.cin) * sizeof(float);
    if (lds > 150 * 1024) return 1;
    int nwg;
    gg_linear_bwd_workspace(p.E, p.cin, p.C, nullptr, &nwg);
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute((const void *)gg_k_linear_bwd<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        hipFuncSetAttribute((const void *)gg_k_linear_bwd<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        hipFuncSetAttribute((const void *)gg_k_linear_bwd<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        hipFuncSetAttribute((const void *)gg_k_linear_bwd<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        hipFuncSetAttribute((const void *)gg_k_linear_bwd<9>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        hipFuncSetAttribute((const void *)gg_k_linear_bwd<12>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_done = true;
    }
    if (pairs_per_wave <= 1) gg_k_linear_bwd<1><<<nwg, 256, lds, st>>>(p);
    else if (pairs_per_wave <= 2) gg_k_linear_bwd<2><<<nwg, 256, lds, st>>>(p);
    else if (pairs_per_wave <= 3) gg_k_linear_bwd<3><<<nwg, 256, lds, st>>>(p);
    else if (pairs_per_wave <= 5) gg_k_linear_bwd<5><<<nwg, 256, lds, st>>>(p);
    else if (pairs_per_wave <= 9) gg_k_linear_bwd<9><<<nwg, 256, lds, st>>>(p);
    else if (pairs_per_wave <= 12) gg_k_linear_bwd<12><<<nwg, 256, lds, st>>>(p);
    else return 1;
    if (hipGetLastError() != hipSuccess) return 3;
    const int cinP = ntm * 32, CP = ntn2 * 32;
    gg_k_dw_reduce<<<(cinP * CP + 255) / 256, 256, 0, st>>>(p.dWpart, nwg, cinP, CP, p.cin, p.C, p.dW);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

// ------------------------------------------------------------------------------------------
// Y = relu(Z*scale[c] + shift[c]);  C % 4 == 0 -> float4
__global__ __launch_bounds__(256) void gg_k_bn_apply(const float *__restrict__ Z,
                                                     const float *__restrict__ scale,
                                                     const float *__restrict__ shift,
                                                     float *__restrict__ Y, long long total, int C)
{
    if ((C & 3) == 0) {
        const long long n4 = total >> 2;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4;
             i += (long long)gridDim.x * 256) {
            float4 z = ((const float4 *)Z)[i];
            const int c = (int)((i * 4) % C);
            const float4 sc = *(const float4 *)(scale + c), sh = *(const float4 *)(shift + c);
            float4 y;
            y.x = fmaxf(z.x * sc.x + sh.x, 0.f); y.y = fmaxf(z.y * sc.y + sh.y, 0.f);
            y.z = fmaxf(z.z * sc.z + sh.z, 0.f); y.w = fmaxf(z.w * sc.w + sh.w, 0.f);
            ((float4 *)Y)[i] = y;
        }
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
             i += (long long)gridDim.x * 256) {
            const int c = (int)(i % C);
            Y[i] = fmaxf(Z[i] * scale[c] + shift[c], 0.f);
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
                                                          double *__restrict__ sums)
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
            const float d = (z * sc + sf > 0.f) ? dY[r * C + c] : 0.f;
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
                const float d = (z * sc + sf > 0.f) ? dY[r * C + c] : 0.f;
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

int gg_bn_apply(const float *Z, const float *scale, const float *shift, float *Y, long long E,
                int C, hipStream_t st)
{
    long long total = E * C;
    gg_k_bn_apply<<<grid_for(total / 4 + 1, 256, 65536), 256, 0, st>>>(Z, scale, shift, Y, total, C);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_bn_bwd_reduce(const float *dY, const float *Z, const float *scale, const float *shift,
                     const float *mean, const float *rstd, long long E, int C, double *sums,
                     hipStream_t st)
{
    if (!((C <= 256 && 256 % C == 0) || (C % 256 == 0))) return 1;
    int rpp = C <= 256 ? 256 / C : 1;
    gg_k_bn_bwd_reduce<<<grid_for(E, rpp * 16, 2048), 256, 0, st>>>(dY, Z, scale, shift, mean, rstd,
                                                                     E, C, sums);
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
