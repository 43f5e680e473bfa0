// gridgcn_knn.hip -- BallKNN / KNN brute-force top-k and the batch_take gather (gfx950).
//
// Replaces BallKNNKernel::Map (gridifyop/ball_k_nn-inl.h:43-94) and KNNKernel::Map
// (k_nn-inl.h:40-92): there every thread streams all `known` points from global memory and
// (KNN) heap-allocates its scratch.  Here a 256-thread block stages tiles of `known` in LDS as
// SoA (all lanes read the same address -> LDS broadcast, no bank conflicts), one query per lane,
// top-k kept in registers with a branch-free stable insertion.
// Arithmetic is pinned: d = ((dx*dx + dy*dy) + dz*dz), fp32, no FMA.
#include "gridgcn_dev.h"
#include <float.h>

#define GG_KNN_TILE 1024

template <int K, bool BALL>
__global__ __launch_bounds__(256) void gg_k_knn(const float *__restrict__ unknown,
                                                const float *__restrict__ known,
                                                const int *__restrict__ downnum,
                                                const int *__restrict__ upnum, int n, int m,
                                                int topk, float r2, int *__restrict__ idx)
{
    __shared__ float sx[GG_KNN_TILE], sy[GG_KNN_TILE], sz[GG_KNN_TILE];
    const int b = blockIdx.y;
    const int qi = blockIdx.x * 256 + threadIdx.x;
    int dn = downnum[b];
    if (dn > m) dn = m;
    const int upn = upnum[b];
    const bool active = qi < n && qi < upn;
    float ux = 0.f, uy = 0.f, uz = 0.f;
    if (active) {
        const float *u = unknown + ((size_t)b * n + qi) * 3;
        ux = u[0]; uy = u[1]; uz = u[2];
    }
    float best[K];
    int besti[K];
#pragma unroll
    for (int l = 0; l < K; l++) { best[l] = FLT_MAX; besti[l] = -1; }

    const float *kb = known + (size_t)b * m * 3;
    for (int t0 = 0; t0 < dn; t0 += GG_KNN_TILE) {
        int tn = dn - t0 < GG_KNN_TILE ? dn - t0 : GG_KNN_TILE;
        __syncthreads();
        for (int j = threadIdx.x; j < tn * 3; j += 256) {
            float val = kb[(size_t)t0 * 3 + j];
            int pidx = j / 3, c = j - pidx * 3;
            (c == 0 ? sx : (c == 1 ? sy : sz))[pidx] = val;
        }
        __syncthreads();
        if (active) {
            for (int kk = 0; kk < tn; kk++) {
                float dx = __fsub_rn(ux, sx[kk]);
                float dy = __fsub_rn(uy, sy[kk]);
                float dz = __fsub_rn(uz, sz[kk]);
                float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)),
                                    __fmul_rn(dz, dz));
                if (BALL && d > r2) continue;            // ball_k_nn-inl.h:77
                if (d < best[K - 1]) {                   // stable insertion, strict < (:78-88)
                    const int id = t0 + kk;
#pragma unroll
                    for (int l = K - 1; l >= 1; l--) {
                        bool cprev = d < best[l - 1];
                        bool ccur = d < best[l];
                        besti[l] = cprev ? besti[l - 1] : (ccur ? id : besti[l]);
                        best[l] = cprev ? best[l - 1] : (ccur ? d : best[l]);
                    }
                    bool c0 = d < best[0];
                    besti[0] = c0 ? id : besti[0];
                    best[0] = c0 ? d : best[0];
                }
            }
        }
    }
    if (active) {
        int *o = idx + ((size_t)b * n + qi) * topk;
#pragma unroll
        for (int l = 0; l < K; l++)
            if (l < topk) o[l] = besti[l];
    }
}

template <bool BALL>
static int gg_launch_knn(const float *unknown, const float *known, const int *downnum,
                         const int *upnum, int B, int n, int m, int k, float r2, int *idx,
                         hipStream_t st)
{
    dim3 grid((n + 255) / 256, B);
#define GG_KNN_CASE(KK)                                                                       \
    gg_k_knn<KK, BALL><<<grid, 256, 0, st>>>(unknown, known, downnum, upnum, n, m, k, r2, idx)
    if (k <= 3) GG_KNN_CASE(3);
    else if (k <= 6) GG_KNN_CASE(6);
    else if (k <= 8) GG_KNN_CASE(8);
    else if (k <= 16) GG_KNN_CASE(16);
    else if (k <= 32) GG_KNN_CASE(32);
    else GG_KNN_CASE(64);
#undef GG_KNN_CASE
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_ball_knn(const float *unknown, const float *known, const int *downnum, const int *upnum,
                int B, int n, int m, int k, float radius, int *idx, hipStream_t st)
{
    return gg_launch_knn<true>(unknown, known, downnum, upnum, B, n, m, k, radius * radius, idx,
                               st);
}
int gg_knn(const float *unknown, const float *known, const int *downnum, const int *upnum, int B,
           int n, int m, int k, int *idx, hipStream_t st)
{
    return gg_launch_knn<false>(unknown, known, downnum, upnum, B, n, m, k, 0.f, idx, st);
}

// ------------------------------------------------------------------------------------------
// batch_take_g (utils/ops.py:78-93): out[b,j,:] = data[clip(index[b,j] + b*N, 0, B*N-1), :]
template <typename VT>
__global__ __launch_bounds__(256) void gg_k_take(const VT *__restrict__ data,
                                                 const int *__restrict__ index, int N, int CV,
                                                 int M, long long rows, long long total,
                                                 VT *__restrict__ out)
{
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total;
         e += (long long)gridDim.x * 256) {
        long long r = e / CV;
        int c = (int)(e - r * CV);
        int b = (int)(r / M);
        long long flat = (long long)index[r] + (long long)b * N;
        flat = flat < 0 ? 0 : (flat > rows - 1 ? rows - 1 : flat);
        out[e] = data[flat * CV + c];
    }
}

__global__ __launch_bounds__(256) void gg_k_take_bwd(const float *__restrict__ gout,
                                                     const int *__restrict__ index, int N, int C,
                                                     int M, long long rows, long long total,
                                                     float *__restrict__ gdata)
{
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total;
         e += (long long)gridDim.x * 256) {
        long long r = e / C;
        int c = (int)(e - r * C);
        int b = (int)(r / M);
        long long flat = (long long)index[r] + (long long)b * N;
        flat = flat < 0 ? 0 : (flat > rows - 1 ? rows - 1 : flat);
        atomicAdd(&gdata[flat * C + c], gout[e]);
    }
}

int gg_batch_take(const float *data, const int *index, int B, int N, int C, int M, float *out,
                  hipStream_t st)
{
    long long rows = (long long)B * N;
    if ((C & 3) == 0 && ((uintptr_t)data & 15) == 0 && ((uintptr_t)out & 15) == 0) {
        long long total = (long long)B * M * (C / 4);
        int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
        if (grid < 1) grid = 1;
        gg_k_take<float4><<<grid, 256, 0, st>>>((const float4 *)data, index, N, C / 4, M, rows,
                                                total, (float4 *)out);
    } else {
        long long total = (long long)B * M * C;
        int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
        if (grid < 1) grid = 1;
        gg_k_take<float><<<grid, 256, 0, st>>>(data, index, N, C, M, rows, total, out);
    }
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_batch_take_backward(const float *gout, const int *index, int B, int N, int C, int M,
                           float *gdata, hipStream_t st)
{
    long long rows = (long long)B * N;
    long long total = (long long)B * M * C;
    int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    if (grid < 1) grid = 1;
    gg_k_take_bwd<<<grid, 256, 0, st>>>(gout, index, N, C, M, rows, total, gdata);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
