// gridgcn_knn.hip -- BallKNN / KNN brute-force top-k and the batch_take gather (gfx950).
//
// Replaces BallKNNKernel::Map (gridifyop/ball_k_nn-inl.h:43-94) and KNNKernel::Map
// (k_nn-inl.h:40-92): there every thread streams all `known` points from global memory and
// (KNN) heap-allocates its scratch.  Here a 256-thread block stages tiles of `known` in LDS as
// SoA (all lanes read the same address -> LDS broadcast, no bank conflicts), one query per lane,
// top-k kept in registers with a branch-free stable insertion.
// Arithmetic is pinned: d = ((dx*dx + dy*dy) + dz*dz), fp32, no FMA.
#include "gridgcn_dev.h"
#include "gridgcn_once.h"
#include <float.h>

#define GG_KNN_TILE 1024

template <int K, bool BALL>
__global__ __launch_bounds__(256) void gg_k_knn(const float *__restrict__ unknown,
                                                const float *__restrict__ known,
                                                const int *__restrict__ downnum,
                                                const int *__restrict__ upnum, int n, int m,
                                                int topk, float r2, int *__restrict__ idx,
                                                int su, int sk, int ztail)
{
    // su / sk: floats per unknown / known point row (3, or the width of the [B,n,4+C] point rows the up
    // path reads in place); ztail: rows >= upnum[b] are written as 0 instead of being left alone
    __shared__ float sx[GG_KNN_TILE], sy[GG_KNN_TILE], sz[GG_KNN_TILE];
    const int b = blockIdx.y;
    const int qi = blockIdx.x * 256 + threadIdx.x;
    int dn = downnum[b];
    if (dn > m) dn = m;
    const int upn = upnum[b];
    const bool active = qi < n && qi < upn;
    float ux = 0.f, uy = 0.f, uz = 0.f;
    if (active) {
        const float *u = unknown + ((size_t)b * n + qi) * su;
        ux = u[0]; uy = u[1]; uz = u[2];
    }
    float best[K];
    int besti[K];
#pragma unroll
    for (int l = 0; l < K; l++) { best[l] = FLT_MAX; besti[l] = -1; }

    const float *kb = known + (size_t)b * m * sk;
    for (int t0 = 0; t0 < dn; t0 += GG_KNN_TILE) {
        int tn = dn - t0 < GG_KNN_TILE ? dn - t0 : GG_KNN_TILE;
        __syncthreads();
        for (int j = threadIdx.x; j < tn * 3; j += 256) {
            int pidx = j / 3, c = j - pidx * 3;
            float val = kb[(size_t)(t0 + pidx) * sk + c];
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
    } else if (ztail && qi < n) {
        for (int l = 0; l < topk; l++) idx[((size_t)b * n + qi) * topk + l] = 0;
    }
}

template <bool BALL>
static int gg_launch_knn(const float *unknown, const float *known, const int *downnum,
                         const int *upnum, int B, int n, int m, int k, float r2, int *idx,
                         hipStream_t st, int su = 3, int sk = 3, int ztail = 0)
{
    dim3 grid((n + 255) / 256, B);
#define GG_KNN_CASE(KK)                                                                       \
    gg_k_knn<KK, BALL><<<grid, 256, 0, st>>>(unknown, known, downnum, upnum, n, m, k, r2, idx, su, sk, ztail)
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
                int B, int n, int m, int k, float radius, int *idx, hipStream_t st, int su, int sk, int ztail)
{
    return gg_launch_knn<true>(unknown, known, downnum, upnum, B, n, m, k, radius * radius, idx,
                               st, su, sk, ztail);
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
                                                     float *__restrict__ gdata, int gs, int ds)
{
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total;
         e += (long long)gridDim.x * 256) {
        long long r = e / C;
        int c = (int)(e - r * C);
        int b = (int)(r / M);
        long long flat = (long long)index[r] + (long long)b * N;
        flat = flat < 0 ? 0 : (flat > rows - 1 ? rows - 1 : flat);
        atomicAdd(&gdata[flat * ds + c], gout[r * gs + c]);
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

// LDS-privatised scatter-add: a workgroup owns (cloud b, channel slice [c0,c0+cs), edge range) and
// accumulates into an LDS copy of the cloud's destination rows (ds_add_f32), then flushes with one
// global atomic per touched element.  At cfg4 up2 (3.3 M edges -> 8192 rows) this replaces 432 M
// memory-side atomics (1.9 ms) by 10 M.  Row N of the LDS tile collects index -1, which the
// reference's clipped flat take sends to row b*N-1 of the flattened batch (utils/ops.py:89-92).
__global__ __launch_bounds__(1024) void gg_k_take_bwd_lds(const float *__restrict__ gout,
                                                          const int *__restrict__ index, int B,
                                                          int N, int C, int M, int cs, int nsplit,
                                                          float *__restrict__ gdata, int gs,
                                                          int ds)
{
    extern __shared__ __attribute__((aligned(16))) float acc[];   // [N+1][cs]
    const int b = blockIdx.z, c0 = blockIdx.y * cs, sp = blockIdx.x;
    const int cw = (c0 + cs <= C) ? cs : C - c0;                  // live channels of this slice
    const int tot = (N + 1) * cs;
    for (int j = threadIdx.x; j < tot; j += 1024) acc[j] = 0.0f;
    __syncthreads();
    const long long rows = (long long)B * N;
    const int per = (M + nsplit - 1) / nsplit;
    const int m0 = sp * per, m1 = (m0 + per < M) ? m0 + per : M;
    const int lanes_c = cs;                                       // power of two <= 1024
    const int epb = 1024 / lanes_c;                               // edges per block iteration
    const int tc = threadIdx.x & (lanes_c - 1), te = threadIdx.x / lanes_c;
    for (int m = m0 + te; m < m1; m += epb) {
        if (tc < cw) {
            long long r = (long long)b * M + m;
            long long flat = (long long)index[r] + (long long)b * N;
            flat = flat < 0 ? 0 : (flat > rows - 1 ? rows - 1 : flat);
            long long li = flat - (long long)b * N;               // -1 .. N-1 (own cloud) normally
            float g = gout[r * gs + c0 + tc];
            if (li >= -1 && li < N) atomicAdd(&acc[(li < 0 ? N : (int)li) * cs + tc], g);
            else atomicAdd(&gdata[flat * ds + c0 + tc], g);        // clipped into another cloud
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < tot; j += 1024) {
        float v = acc[j];
        if (v != 0.0f) {
            int row = j / cs, c = j - row * cs;
            if (c < cw) {
                long long flat = (row == N) ? ((long long)b * N - 1) : ((long long)b * N + row);
                if (flat < 0) flat = 0;
                atomicAdd(&gdata[flat * ds + c0 + c], v);
            }
        }
    }
}

// gout row stride gs, gdata row stride ds (floats); C channels are added
int gg_batch_take_backward(const float *gout, const int *index, int B, int N, int C, int M,
                           float *gdata, int gs, int ds, hipStream_t st)
{
    // LDS path when a useful channel slice of the whole cloud fits in 144 KB of LDS
    int cs = 1;
    while (cs < 128 && cs < C) cs <<= 1;
    while (cs > 1 && (size_t)(N + 1) * cs * 4 > 144 * 1024) cs >>= 1;
    if (cs >= 8 && (size_t)(N + 1) * cs * 4 <= 144 * 1024 && (long long)M >= 4LL * N) {
        static GGDevOnce attr_done;
        if (!attr_done) {
            if (hipFuncSetAttribute((const void *)gg_k_take_bwd_lds,
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    144 * 1024) != hipSuccess) return 3;
            attr_done = true;
        }
        int nslice = (C + cs - 1) / cs;
        int nsplit = 512 / (B * nslice);
        if (nsplit < 1) nsplit = 1;
        if (nsplit > (M + 4095) / 4096) nsplit = (M + 4095) / 4096;
        gg_k_take_bwd_lds<<<dim3(nsplit, nslice, B), 1024, (size_t)(N + 1) * cs * 4, st>>>(
            gout, index, B, N, C, M, cs, nsplit, gdata, gs, ds);
        return hipGetLastError() == hipSuccess ? 0 : 3;
    }
    long long rows = (long long)B * N;
    long long total = (long long)B * M * C;
    int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    if (grid < 1) grid = 1;
    gg_k_take_bwd<<<grid, 256, 0, st>>>(gout, index, N, C, M, rows, total, gdata, gs, ds);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

// ------------------------------------------------------------------------------------------
// edge inputs of sub_g_update (gcn_module_g_att.py:190-194, 217-218, 242-250) in ONE pass:
//   nf  [E, cin] = geo_vec | features | concat(geo_vec, features)
//   att [E, 10]  = (geo_dist, geo_vec, centre xyz, neighbour xyz)
// replaces batch_take_g + slice_axis + tile + elemwise_sub + sqrt(sum(square)) + 2 concats.
__global__ __launch_bounds__(256) void gg_k_edge_inputs(const float *__restrict__ src,
                                                        const int *__restrict__ nebidx,
                                                        const float *__restrict__ cent,
                                                        int cent_stride, int B, int Nsrc, int Cs,
                                                        int O, int P, int has_feats, int geo,
                                                        int cin, long long total,
                                                        float *__restrict__ nf,
                                                        float *__restrict__ att)
{
    const long long rows = (long long)B * Nsrc;
    const int w = cin + 10;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total;
         t += (long long)gridDim.x * 256) {
        long long e = t / w;
        int c = (int)(t - e * w);
        long long ci = e / P;
        int b = (int)(ci / O);
        long long flat = (long long)nebidx[e] + (long long)b * Nsrc;
        flat = flat < 0 ? 0 : (flat > rows - 1 ? rows - 1 : flat);
        const float *srow = src + flat * Cs;
        const int fo = geo ? 3 : 0;
        if (c < cin && c >= fo) {
            nf[e * cin + c] = srow[4 + c - fo];
        } else {
            const float *cen = cent + ci * cent_stride;
            float cx = cen[0], cy = cen[1], cz = cen[2];
            float nx = srow[0], ny = srow[1], nz = srow[2];
            float gx = nx - cx, gy = ny - cy, gz = nz - cz;
            if (c < cin) {
                nf[e * cin + c] = c == 0 ? gx : (c == 1 ? gy : gz);
            } else {
                int a = c - cin;
                float v;
                switch (a) {
                case 0: v = sqrtf((gx * gx + gy * gy) + gz * gz); break;
                case 1: v = gx; break; case 2: v = gy; break; case 3: v = gz; break;
                case 4: v = cx; break; case 5: v = cy; break; case 6: v = cz; break;
                case 7: v = nx; break; case 8: v = ny; break; default: v = nz; break;
                }
                att[e * 10 + a] = v;
            }
        }
    }
}

// "features first" layout for the MFMA kernels: one edge per half-wave, 16-byte loads and stores.
//   nf [E, nfs]  = features (nfeat = Cs-4, a multiple of 4) | geo_vec (if geo) | zeros up to nfs
//   att[E, 16]   = (geo_dist, geo_vec, centre xyz, neighbour xyz, 0 x 6)
// nfs is a multiple of 8 so that every row is a whole number of 32-byte half-lines
// (gg_k_linear_fwd_direct reads rows with 16-byte loads; the matching permutation of the weight
// columns is done by gridgcn_pack_linear).
__global__ __launch_bounds__(256) void gg_k_edge_inputs_rows(
    const float *__restrict__ src, const int *__restrict__ nebidx, const float *__restrict__ cent,
    int cent_stride, int Nsrc, long long rows, int Cs, int O, int P, int nfeat, int geo, int nfs,
    int E, float *__restrict__ nf, float *__restrict__ att)
{
    const int sub = threadIdx.x & 31;
    const int nhalf = gridDim.x * 8;
    const int ntail4 = (nfs - nfeat) >> 2;
    for (int e = blockIdx.x * 8 + (threadIdx.x >> 5); e < E; e += nhalf) {
        const int ci = e / P;
        const int b = ci / O;
        long long flat = (long long)nebidx[e] + (long long)b * Nsrc;
        flat = flat < 0 ? 0 : (flat > rows - 1 ? rows - 1 : flat);
        const float *srow = src + flat * Cs;
        float *nrow = nf + (size_t)e * nfs;
        for (int i = sub * 4; i < nfeat; i += 128)
            *(float4 *)(nrow + i) = *(const float4 *)(srow + 4 + i);
        // the last lanes of the half-wave write the geo tail and the attention row
        const int j = 31 - sub;                      // 0..3: att float4 j; 4..: tail float4 j-4
        if (j < 4 + ntail4) {
            const float *cen = cent + (size_t)ci * cent_stride;
            const float cx = cen[0], cy = cen[1], cz = cen[2];
            const float nx = srow[0], ny = srow[1], nz = srow[2];
            const float gx = nx - cx, gy = ny - cy, gz = nz - cz;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j == 0) v = make_float4(sqrtf((gx * gx + gy * gy) + gz * gz), gx, gy, gz);
            else if (j == 1) v = make_float4(cx, cy, cz, nx);
            else if (j == 2) v = make_float4(ny, nz, 0.f, 0.f);
            else if (j == 4 && geo) v = make_float4(gx, gy, gz, 0.f);
            if (j < 4) *(float4 *)(att + (size_t)e * 16 + 4 * j) = v;
            else *(float4 *)(nrow + nfeat + 4 * (j - 4)) = v;
        }
    }
}

// the same rows for a layer WITHOUT neighbour features (nfeat = 0: the first down layer, whose source
// rows are the raw points): nothing to copy, so one edge per lane instead of one per half-wave -- the
// half-wave form left 26 of 32 lanes idle and ran at a third of the bandwidth of its 96 bytes/edge
__global__ __launch_bounds__(256) void gg_k_edge_inputs_rows_geo(
    const float *__restrict__ src, const int *__restrict__ nebidx, const float *__restrict__ cent,
    int cent_stride, int Nsrc, long long rows, int Cs, int O, int P, int geo, int nfs, int E,
    float *__restrict__ nf, float *__restrict__ att)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const int ci = e / P;
    const int b = ci / O;
    long long flat = (long long)nebidx[e] + (long long)b * Nsrc;
    flat = flat < 0 ? 0 : (flat > rows - 1 ? rows - 1 : flat);
    const float *srow = src + flat * Cs;
    const float *cen = cent + (size_t)ci * cent_stride;
    const float cx = cen[0], cy = cen[1], cz = cen[2];
    const float nx = srow[0], ny = srow[1], nz = srow[2];
    const float gx = nx - cx, gy = ny - cy, gz = nz - cz;
    float4 *a = (float4 *)(att + (size_t)e * 16);
    a[0] = make_float4(sqrtf((gx * gx + gy * gy) + gz * gz), gx, gy, gz);
    a[1] = make_float4(cx, cy, cz, nx);
    a[2] = make_float4(ny, nz, 0.f, 0.f);
    a[3] = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 *n = (float4 *)(nf + (size_t)e * nfs);
    n[0] = geo ? make_float4(gx, gy, gz, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 1; j < (nfs >> 2); j++) n[j] = make_float4(0.f, 0.f, 0.f, 0.f);
}

int gg_edge_inputs_rows(const float *src, const int *nebidx, const float *cent, int cent_stride,
                        int B, int Nsrc, int Cs, int O, int P, int has_feats, int localfdim,
                        int nfs, float *nf, float *att, hipStream_t st)
{
    const int geo = (!has_feats || localfdim != 0) ? 1 : 0;
    const int nfeat = has_feats ? Cs - 4 : 0;
    if ((nfeat & 3) || (nfs & 7) || nfs < nfeat + (geo ? 3 : 0) || nfs - nfeat > 96) return 1;
    const long long E = (long long)B * O * P;
    if (E >= (1ll << 31)) return 1;
    if (nfeat == 0) {
        gg_k_edge_inputs_rows_geo<<<(int)((E + 255) / 256), 256, 0, st>>>(
            src, nebidx, cent, cent_stride, Nsrc, (long long)B * Nsrc, Cs, O, P, geo, nfs, (int)E, nf,
            att);
        return hipGetLastError() == hipSuccess ? 0 : 3;
    }
    long long nb = (E + 7) / 8;
    const int grid = (int)(nb < 65536 ? nb : 65536);
    gg_k_edge_inputs_rows<<<grid, 256, 0, st>>>(src, nebidx, cent, cent_stride, Nsrc,
                                                (long long)B * Nsrc, Cs, O, P, nfeat, geo, nfs,
                                                (int)E, nf, att);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_edge_inputs(const float *src, const int *nebidx, const float *cent, int cent_stride, int B,
                   int Nsrc, int Cs, int O, int P, int has_feats, int localfdim, float *nf,
                   float *att, hipStream_t st)
{
    const int geo = (!has_feats || localfdim != 0) ? 1 : 0;
    const int cin = (geo ? 3 : 0) + (has_feats ? Cs - 4 : 0);
    long long total = (long long)B * O * P * (cin + 10);
    long long nb = (total + 255) / 256;
    int grid = (int)(nb < 262144 ? nb : 262144);
    gg_k_edge_inputs<<<grid, 256, 0, st>>>(src, nebidx, cent, cent_stride, B, Nsrc, Cs, O, P,
                                           has_feats, geo, cin, total, nf, att);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
