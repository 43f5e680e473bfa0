// gridgcn_scatter.hip -- backward of the neighbour gather (batch_take_g, utils/ops.py:78-93) as a
// SORTED segmented sum instead of a scatter-add (gfx950).
//
// grad_data[clip(index[b,m] + b*N), :] += grad_out[b, m, :] touches each destination row dozens of
// times; with atomics (even LDS-privatised ones, gridgcn_knn.hip) the [E, C] gradient is read in
// 32-byte slivers and the adds serialise (2.1 ms for the 3.3 M-edge layer of cfg4).  Here the edges
// of a cloud are first ordered by destination row (a counting sort on the index tensor alone, ~4 MB
// of traffic), then every wave walks a fixed chunk of 256 sorted edges, reading whole gradient rows
// (512 contiguous bytes) and summing runs of equal destination in registers:
//   gg_k_csr_hist     per (cloud, part): LDS histogram of the destination keys
//   gg_k_csr_scan     per cloud: row offsets + the base offset of every (part, key)
//   gg_k_csr_scatter  per (cloud, part): perm[] / keys[] in sorted order (LDS cursors)
//   gg_k_take_bwd_sorted  the segmented sum; a run that is a complete row is stored, a run cut by a
//                     chunk boundary is added atomically (<= 2 per chunk)
// key of an edge = flat row - (b*N - 1) in [0, N]: key 0 is the reference's clipped "-1 -> last row
// of the previous cloud" (mx.sym.take mode='clip' on the flattened batch), so key 0 of cloud b and
// key N of cloud b-1 share a destination and are always added atomically.  key N+1 collects
// indices clipped into any other cloud (never produced by the index ops; handled edge by edge).
// The order of the edges inside a run depends on the LDS atomics of the sort: sums are reproducible
// to fp32 round-off, not bit for bit (as the framework's own scatter-add backward).
#include "gridgcn_csr.h"
#include "gridgcn_once.h"

__device__ __forceinline__ int gg_csr_key(int idx, int b, int N, long long rows)
{
    long long flat = (long long)idx + (long long)b * N;
    flat = flat < 0 ? 0 : (flat > rows - 1 ? rows - 1 : flat);
    const long long li = flat - ((long long)b * N - 1);
    return (li < 0 || li > N) ? N + 1 : (int)li;
}

// hist[b][part][N+2]
__global__ __launch_bounds__(1024) void gg_k_csr_hist(const int *__restrict__ index, int B, int N,
                                                      int M, int *__restrict__ hist)
{
    extern __shared__ int cnt[];
    const int b = blockIdx.y, part = blockIdx.x, nk = N + 2;
    for (int j = threadIdx.x; j < nk; j += 1024) cnt[j] = 0;
    __syncthreads();
    const long long rows = (long long)B * N;
    const int per = (M + GG_CSR_PARTS - 1) / GG_CSR_PARTS;
    const int m0 = part * per, m1 = m0 + per < M ? m0 + per : M;
    // four index loads in flight per thread (one by one: a memory round trip per 1024 edges); past the end the last
    // edge is read again and counted with 0 -- no branch around the use, or the load is sunk into it
    for (int m = m0 + threadIdx.x; m < m1; m += 4096) {
        int v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = index[(size_t)b * M + (m + 1024 * u < m1 ? m + 1024 * u : m1 - 1)];
#pragma unroll
        for (int u = 0; u < 4; u++) atomicAdd(&cnt[gg_csr_key(v[u], b, N, rows)], m + 1024 * u < m1 ? 1 : 0);
    }
    __syncthreads();
    int *out = hist + ((size_t)b * GG_CSR_PARTS + part) * nk;
    for (int j = threadIdx.x; j < nk; j += 1024) out[j] = cnt[j];
}

// rowptr[b][N+3] (exclusive offsets, last = M); hist is rewritten to base offsets per (part, key)
__global__ __launch_bounds__(1024) void gg_k_csr_scan(int N, int *__restrict__ hist,
                                                      int *__restrict__ rowptr)
{
    __shared__ int wsum[16];
    __shared__ int carry;
    const int b = blockIdx.x, nk = N + 2;
    int *h = hist + (size_t)b * GG_CSR_PARTS * nk;
    int *rp = rowptr + (size_t)b * (N + 3);
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k0 = 0; k0 < nk; k0 += 1024) {
        const int k = k0 + threadIdx.x;
        int tot = 0;
        if (k < nk)
            {
                int c[GG_CSR_PARTS];                       // (the 16 loads together, added in part order)
#pragma unroll
                for (int s = 0; s < GG_CSR_PARTS; s++) c[s] = h[(size_t)s * nk + k];
#pragma unroll
                for (int s = 0; s < GG_CSR_PARTS; s++) tot += c[s];
            }
        // inclusive scan of tot over the block
        int v = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(v, o, 64);
            if (lane >= o) v += t;
        }
        if (lane == 63) wsum[wave] = v;
        __syncthreads();
        int base = carry;
        for (int w = 0; w < wave; w++) base += wsum[w];
        const int excl = base + v - tot;
        if (k < nk) {
            rp[k] = excl;
            int run = excl;
            int c[GG_CSR_PARTS];
#pragma unroll
            for (int s = 0; s < GG_CSR_PARTS; s++) c[s] = h[(size_t)s * nk + k];
#pragma unroll
            for (int s = 0; s < GG_CSR_PARTS; s++) {
                h[(size_t)s * nk + k] = run;
                run += c[s];
            }
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry = base + v;
        __syncthreads();
    }
    if (threadIdx.x == 0) rp[nk] = carry;
}

__global__ __launch_bounds__(1024) void gg_k_csr_scatter(const int *__restrict__ index, int B,
                                                         int N, int M, const int *__restrict__ hist,
                                                         int *__restrict__ perm,
                                                         int *__restrict__ keys)
{
    extern __shared__ int cur[];
    const int b = blockIdx.y, part = blockIdx.x, nk = N + 2;
    const int *base = hist + ((size_t)b * GG_CSR_PARTS + part) * nk;
    for (int j = threadIdx.x; j < nk; j += 1024) cur[j] = base[j];
    __syncthreads();
    const long long rows = (long long)B * N;
    const int per = (M + GG_CSR_PARTS - 1) / GG_CSR_PARTS;
    const int m0 = part * per, m1 = m0 + per < M ? m0 + per : M;
    for (int m = m0 + threadIdx.x; m < m1; m += 4096) {       // (four index loads in flight, as in gg_k_csr_hist)
        int v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = index[(size_t)b * M + (m + 1024 * u < m1 ? m + 1024 * u : m1 - 1)];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool ok = m + 1024 * u < m1;
            const int key = gg_csr_key(v[u], b, N, rows);
            const int pos = atomicAdd(&cur[key], ok ? 1 : 0);
            if (ok) {
                perm[(size_t)b * M + pos] = m + 1024 * u;
                keys[(size_t)b * M + pos] = key;
            }
        }
    }
}

template <int VPL> struct GGVec;
template <> struct GGVec<1> { typedef float T; };
template <> struct GGVec<2> { typedef float2 T; };
template <> struct GGVec<4> { typedef float4 T; };

// one wave per chunk of GG_CSR_CHUNK sorted edges; lane l owns channels [l*VPL, l*VPL+VPL)
template <int VPL>
__global__ __launch_bounds__(256) void gg_k_take_bwd_sorted(
    const float *__restrict__ gout, int gs, const int *__restrict__ perm,
    const int *__restrict__ keys, const int *__restrict__ rowptr, const int *__restrict__ index,
    int B, int N, int C, int M, int cpc, float *__restrict__ gdata, int ds)
{
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (wid >= B * cpc) return;
    const int b = wid / cpc, ch = wid - b * cpc;
    const int e0 = ch * GG_CSR_CHUNK, e1 = e0 + GG_CSR_CHUNK < M ? e0 + GG_CSR_CHUNK : M;
    const int c = lane * VPL;
    const bool live = c < C;
    const long long rows = (long long)B * N;
    const int *pk = keys + (size_t)b * M, *pp = perm + (size_t)b * M;
    const int *rp = rowptr + (size_t)b * (N + 3);
    const float *g0 = gout + (size_t)b * M * gs + (live ? c : 0);
    float acc[VPL];
#pragma unroll
    for (int i = 0; i < VPL; i++) acc[i] = 0.f;
    int cur = pk[e0], rs = e0;

    auto flush = [&](int key, int rbeg, int rend) {
        if (key > N || !live) return;
        long long dest = (long long)b * N - 1 + key;
        if (dest < 0) dest = 0;
        float *d = gdata + dest * ds + c;
        const bool whole = key != 0 && key != N && rbeg == rp[key] && rend == rp[key + 1];
        if (whole) {
#pragma unroll
            for (int i = 0; i < VPL; i++) d[i] = acc[i];
        } else {
#pragma unroll
            for (int i = 0; i < VPL; i++) atomicAdd(&d[i], acc[i]);
        }
    };

    constexpr int G = 8;
    for (int e = e0; e < e1; e += G) {
        int k[G], m[G];
        typename GGVec<VPL>::T v[G];
#pragma unroll
        for (int j = 0; j < G; j++) {
            const int ee = e + j < e1 ? e + j : e1 - 1;
            k[j] = pk[ee];
            m[j] = pp[ee];
        }
#pragma unroll
        for (int j = 0; j < G; j++)
            v[j] = *(const typename GGVec<VPL>::T *)(g0 + (size_t)m[j] * gs);
#pragma unroll
        for (int j = 0; j < G; j++) {
            if (e + j >= e1) break;
            const float *vf = (const float *)&v[j];
            if (k[j] == N + 1) {
                // clipped into a foreign cloud: one atomic row add per edge
                if (cur != N + 1) { flush(cur, rs, e + j); cur = N + 1; }
                if (live) {
                    long long flat = (long long)index[(size_t)b * M + m[j]] + (long long)b * N;
                    flat = flat < 0 ? 0 : (flat > rows - 1 ? rows - 1 : flat);
#pragma unroll
                    for (int i = 0; i < VPL; i++) atomicAdd(&gdata[flat * ds + c + i], vf[i]);
                }
                continue;
            }
            if (k[j] != cur) {
                flush(cur, rs, e + j);
                cur = k[j];
                rs = e + j;
#pragma unroll
                for (int i = 0; i < VPL; i++) acc[i] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < VPL; i++) acc[i] += vf[i];
        }
    }
    flush(cur, rs, e1);
}

size_t gg_csr_workspace(int B, int N, int M)
{
    return ((size_t)2 * B * M + (size_t)B * (N + 3) + (size_t)B * GG_CSR_PARTS * (N + 2)) * sizeof(int);
}

int gg_csr_build(const int *index, int B, int N, int M, void *workspace, int **perm_, int **keys_,
                 int **rowptr_, hipStream_t st)
{
    if ((size_t)(N + 2) * 4 > 150 * 1024 || M < 1 || B < 1) return 1;
    int *perm = (int *)workspace;
    int *keys = perm + (size_t)B * M;
    int *rowptr = keys + (size_t)B * M;
    int *hist = rowptr + (size_t)B * (N + 3);
    static GGDevOnce attr_done;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)gg_k_csr_hist, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 3;
        if (hipFuncSetAttribute((const void *)gg_k_csr_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 3;
        attr_done = true;
    }
    const size_t lds = (size_t)(N + 2) * 4;
    gg_k_csr_hist<<<dim3(GG_CSR_PARTS, B), 1024, lds, st>>>(index, B, N, M, hist);
    gg_k_csr_scan<<<B, 1024, 0, st>>>(N, hist, rowptr);
    gg_k_csr_scatter<<<dim3(GG_CSR_PARTS, B), 1024, lds, st>>>(index, B, N, M, hist, perm, keys);
    *perm_ = perm; *keys_ = keys; *rowptr_ = rowptr;
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

size_t gg_take_bwd_sorted_workspace(int B, int N, int M) { return gg_csr_workspace(B, N, M); }

// 1 = shape not supported (caller falls back to the scatter-add kernels)
int gg_take_bwd_sorted(const float *gout, const int *index, int B, int N, int C, int M, float *gdata,
                       int gs, int ds, void *workspace, hipStream_t st)
{
    if (C < 1 || C > 256) return 1;
    const int VPL = C <= 64 ? 1 : (C <= 128 ? 2 : 4);
    if ((gs % VPL) || (ds % VPL) || (C % VPL) || ((uintptr_t)gout & (4 * VPL - 1)) ||
        ((uintptr_t)gdata & (4 * VPL - 1)))
        return 1;
    int *perm, *keys, *rowptr;
    const int rc = gg_csr_build(index, B, N, M, workspace, &perm, &keys, &rowptr, st);
    if (rc) return rc;
    const int cpc = (M + GG_CSR_CHUNK - 1) / GG_CSR_CHUNK;
    const int nwave = B * cpc, grid = (nwave + 3) / 4;
    if (VPL == 1) gg_k_take_bwd_sorted<1><<<grid, 256, 0, st>>>(gout, gs, perm, keys, rowptr, index, B, N, C, M, cpc, gdata, ds);
    else if (VPL == 2) gg_k_take_bwd_sorted<2><<<grid, 256, 0, st>>>(gout, gs, perm, keys, rowptr, index, B, N, C, M, cpc, gdata, ds);
    else gg_k_take_bwd_sorted<4><<<grid, 256, 0, st>>>(gout, gs, perm, keys, rowptr, index, B, N, C, M, cpc, gdata, ds);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
