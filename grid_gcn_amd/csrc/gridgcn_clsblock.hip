// gridgcn_clsblock.hip -- the pieces of the CLASSIFICATION GridConv edge block that the
// segmentation block does not have (classification/models/gcn_module_g.py:64-114 verts_pair_func
// with att_full='next', :212-223 contextvec_func), gfx950.
//
// The attention MLP of the classifier reads  concat(att1(att_vec), pt_mlp(nf), context)  per edge,
// where context = max over the centre's P neighbours of the raw edge features, tiled back over the
// neighbours.  None of the three pieces is concatenated here: the first attention conv is
//     Z = act(Z_att1) Wa^T + act(Z_pt) Wn^T + (ctx Wc^T + b)[centre]
// i.e. the two-source register-direct GEMM of gridgcn_direct.hip (GGLinFwd::X2) with a bias per
// centre (GGLinFwd::rowbias).  This file holds what surrounds it:
//   gg_k_ctx_max      ctx[centre] = max_p (geo_vec | neighbour features) straight from the source
//                     points (+ the arg-max source row per feature column, for the backward)
//   gg_k_ctx_scatter  d src[arg-max row] += d ctx
//   gg_k_dz_segsum    d(bias per centre) = sum_p dZ with dZ = BatchNorm/ReLU backward of (dY, Z)
//   gg_k_sparse_add   dense gradient += the sparse (arg-max, value) gradient of the product/max
#include "gridgcn_dev.h"
#include "gridgcn_clsblock.h"

// one workgroup per centre, thread j = column j of the context vector (3 geo columns, then Cf
// feature columns): rows of the P neighbours are read coalesced over j.
__global__ __launch_bounds__(256) void gg_k_ctx_max(const float *__restrict__ src,
                                                    const int *__restrict__ nebidx,
                                                    const float *__restrict__ cent, int cent_stride,
                                                    int B, int Nsrc, int Cs, int O, int P,
                                                    float *__restrict__ ctx, int *__restrict__ cidx)
{
    __shared__ int sidx[256];
    const int ci = blockIdx.x, bi = ci / O, Cf = Cs - 4, cin = 3 + Cf;
    const long long rows = (long long)B * Nsrc;
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        long long flat = (long long)nebidx[(size_t)ci * P + p] + (long long)bi * Nsrc;
        flat = flat < 0 ? 0 : (flat > rows - 1 ? rows - 1 : flat);      // take(mode='clip')
        sidx[p] = (int)flat;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < cin; j += blockDim.x) {
        const int col = j < 3 ? j : j + 1;                               // skip the weight column
        const float c0 = j < 3 ? cent[(size_t)ci * cent_stride + j] : 0.f;
        float best = -INFINITY;
        int bi_ = 0;
        // eight gathered rows in flight, compared in neighbour order (the first maximum wins, as one by one: a load per
        // iteration behind a data-dependent branch was P memory round trips in a row -- tools/isa_chains.py)
        int p = 0;
        for (; p + 8 <= P; p += 8) {
            int ri[8];
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) ri[u] = sidx[p + u];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = src[(size_t)ri[u] * Cs + col];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const float d = v[u] - c0;
                const bool gt = d > best;
                best = gt ? d : best;
                bi_ = gt ? ri[u] : bi_;
            }
        }
        for (; p < P; p++) {
            const float v = src[(size_t)sidx[p] * Cs + col] - c0;
            if (v > best) { best = v; bi_ = sidx[p]; }
        }
        ctx[(size_t)ci * cin + j] = best;
        if (j >= 3 && cidx) cidx[(size_t)ci * Cf + (j - 3)] = bi_;
    }
}

// dsrc [B*Nsrc][Cs] += : feature column j of the arg-max row receives dctx[centre][3 + j]
__global__ __launch_bounds__(256) void gg_k_ctx_scatter(const float *__restrict__ dctx,
                                                        const int *__restrict__ cidx,
                                                        long long ncent, int Cf, int Cs,
                                                        float *__restrict__ dsrc)
{
    const long long total = ncent * Cf;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long ci = i / Cf;
        const int j = (int)(i - ci * Cf);
        const float g = dctx[ci * (3 + Cf) + 3 + j];
        if (g != 0.f) atomicAdd(&dsrc[(size_t)cidx[i] * Cs + 4 + j], g);
    }
}

// out[centre][c] = sum_p dZ[centre*P + p][c],
//   dZ = scale*dyr - scale*m1 - scale*rstd*m2*(z - mean),  dyr = dY * (z*scale + shift > 0)
// (the same expression the dX / dW kernels form in registers).  One workgroup per centre and
// column block of 256; threads of a wave read 256 contiguous bytes of a row.
__global__ __launch_bounds__(256) void gg_k_dz_segsum(const float *__restrict__ dY,
                                                      const float *__restrict__ Z,
                                                      const float *__restrict__ scale,
                                                      const float *__restrict__ shift,
                                                      const float *__restrict__ mean,
                                                      const float *__restrict__ rstd,
                                                      const float *__restrict__ m1,
                                                      const float *__restrict__ m2, int P, int C,
                                                      float *__restrict__ out)
{
    __shared__ float part[256];
    const long long ci = blockIdx.x;
    // C >= 256: one column per thread; narrower: 256/C row streams share the block
    const int rs = C >= 256 ? 1 : 256 / C;
    const int c = C >= 256 ? blockIdx.y * 256 + threadIdx.x : threadIdx.x % C;
    const int r = C >= 256 ? 0 : threadIdx.x / C;
    float s1 = 0.f, sz = 0.f;
    float sc = 0.f, sh = 0.f;
    const bool ok = c < C && r < rs;
    if (ok) {
        sc = scale[c]; sh = shift[c];
        const float *zr = Z + (size_t)ci * P * C + c, *gr = dY + (size_t)ci * P * C + c;
        // eight rows of Z and of dY in flight, added in row order (the sums of the one-by-one loop, bit for bit)
        int p = r;
        for (; p + 7 * rs < P; p += 8 * rs) {
            float z[8], g[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                z[u] = zr[(size_t)(p + u * rs) * C];
                g[u] = gr[(size_t)(p + u * rs) * C];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                s1 += (z[u] * sc + sh > 0.f) ? g[u] : 0.f;
                sz += z[u];
            }
        }
        for (; p < P; p += rs) {
            const float z = zr[(size_t)p * C];
            const float g = gr[(size_t)p * C];
            s1 += (z * sc + sh > 0.f) ? g : 0.f;
            sz += z;
        }
    }
    if (rs > 1) {
        // combine the row streams: two passes through LDS
        part[threadIdx.x] = s1;
        __syncthreads();
        if (r == 0 && ok) for (int k = 1; k < rs; k++) s1 += part[k * C + c];
        __syncthreads();
        part[threadIdx.x] = sz;
        __syncthreads();
        if (r == 0 && ok) for (int k = 1; k < rs; k++) sz += part[k * C + c];
    }
    if (ok && r == 0) {
        const float mu = mean[c];
        out[(size_t)ci * C + c] = sc * s1 - (float)P * sc * m1[c] -
                                  sc * rstd[c] * m2[c] * (sz - (float)P * mu);
    }
}

// dX[(centre*P + amax[centre][c])][c] += gval[centre][c]   (every destination distinct: plain RMW)
__global__ __launch_bounds__(256) void gg_k_sparse_add(const unsigned char *__restrict__ amax,
                                                       const float *__restrict__ gval,
                                                       long long total, int P, int C,
                                                       float *__restrict__ dX)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long ci = i / C;
        const int c = (int)(i - ci * C);
        dX[(size_t)(ci * P + amax[i]) * C + c] += gval[i];
    }
}

// sums[c] += sum_e Z[e][c], sums[C + c] += sum_e Z[e][c]^2 over E rows of ld floats: the batch
// statistics of a layer whose GEMM ran outside the MFMA kernels (their epilogues do this for free).
// block = 256 threads = 256/Cb row streams x Cb columns (Cb = min(C, 256) columns per block column).
__global__ __launch_bounds__(256) void gg_k_bn_stats(const float *__restrict__ Z, long long E, int C,
                                                     int ld, double *__restrict__ sums)
{
    __shared__ float sh1[256], sh2[256];
    const int tid = threadIdx.x;
    const int Cb = C < 256 ? C : 256;
    const int rs = 256 / Cb;
    const int c = blockIdx.y * 256 + tid % Cb, rr = tid / Cb;
    float a1 = 0.f, a2 = 0.f;
    const bool ok = c < C && rr < rs;
    if (ok)
        for (long long r = (long long)blockIdx.x * rs + rr; r < E; r += (long long)gridDim.x * rs) {
            const float z = Z[r * ld + c];
            a1 += z;
            a2 += z * z;
        }
    sh1[tid] = a1; sh2[tid] = a2;
    __syncthreads();
    if (ok && rr == 0) {
        for (int j = 1; j < rs; j++) { a1 += sh1[tid + j * Cb]; a2 += sh2[tid + j * Cb]; }
        atomicAdd(&sums[c], (double)a1);
        atomicAdd(&sums[C + c], (double)a2);
    }
}

static int gg_grid(long long work, int per_block, int cap)
{
    long long nb = (work + per_block - 1) / per_block;
    return (int)(nb < 1 ? 1 : (nb > cap ? cap : nb));
}

int gg_ctx_max(const float *src, const int *nebidx, const float *cent, int cent_stride, int B,
               int Nsrc, int Cs, int O, int P, float *ctx, int *cidx, hipStream_t st)
{
    if (P > 256 || Cs < 4) return 1;
    gg_k_ctx_max<<<B * O, 256, 0, st>>>(src, nebidx, cent, cent_stride, B, Nsrc, Cs, O, P, ctx, cidx);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_ctx_scatter(const float *dctx, const int *cidx, long long ncent, int Cf, int Cs, float *dsrc,
                   hipStream_t st)
{
    if (Cf < 1) return 0;
    gg_k_ctx_scatter<<<gg_grid(ncent * Cf, 256, 16384), 256, 0, st>>>(dctx, cidx, ncent, Cf, Cs, dsrc);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_dz_segsum(const float *dY, const float *Z, const float *scale, const float *shift,
                 const float *mean, const float *rstd, const float *m1, const float *m2,
                 long long ncent, int P, int C, float *out, hipStream_t st)
{
    if (C < 1 || ncent > 0x7fffffffLL) return 1;
    dim3 grid((unsigned)ncent, C >= 256 ? (C + 255) / 256 : 1);
    gg_k_dz_segsum<<<grid, 256, 0, st>>>(dY, Z, scale, shift, mean, rstd, m1, m2, P, C, out);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_sparse_add(const unsigned char *amax, const float *gval, long long ncent, int P, int C, float *dX,
                  hipStream_t st)
{
    gg_k_sparse_add<<<gg_grid(ncent * C, 256, 16384), 256, 0, st>>>(amax, gval, ncent * C, P, C, dX);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_bn_stats(const float *Z, long long E, int C, int ld, double *sums, hipStream_t st)
{
    const int Cb = C < 256 ? C : 256, rs = 256 / Cb;
    dim3 grid((unsigned)gg_grid(E, rs * 32, 2048), (unsigned)((C + 255) / 256));
    gg_k_bn_stats<<<grid, 256, 0, st>>>(Z, E, C, ld, sums);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
