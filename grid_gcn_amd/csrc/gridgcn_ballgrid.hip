// gridgcn_ballgrid.hip -- BallKNN through a uniform cell grid over the known points (gfx950).
//
// BallKNNKernel::Map (gridifyop/ball_k_nn-inl.h:45-93) tests every known point of the cloud for
// every unknown point (cfg4 up2: 81920 x 1024 pairs per cloud, 0.32 ms brute force in LDS tiles,
// gridgcn_knn.hip).  The radius is a few percent of the cloud's extent, so a grid whose cell is
// >= 1.001 * radius leaves ~27 cells x a fraction of a point to test:
//   gg_k_ball_grid_build  one workgroup per cloud: bounding box of the finite known points, cell
//                         size, counting sort of the points into cells (LDS atomics), cellStart[]
//   gg_k_ball_grid_query  one thread per unknown point: the 27 neighbouring cells (nine runs of the
//                         sorted records, which carry the coordinates), exact distance (same fp32
//                         expression), top-k by (distance, index)
// Identical output to the sequential scan: the reference inserts with a strict `<`, i.e. it keeps
// the k smallest by (distance, index) -- an order that does not depend on the traversal.  A point
// within the radius differs by less than one cell per axis, and the cell function is monotone, so
// it is always in the 3x3x3 neighbourhood (also after clamping to the box).  Non-finite known
// points can never be within the radius (their distance is inf or NaN) and are left out.
#include <hip/hip_runtime.h>
#include <float.h>

#define GG_BG_DMAX 24
#define GG_BG_NCMAX (GG_BG_DMAX * GG_BG_DMAX * GG_BG_DMAX)

struct GGBallGridInfo { float ox, oy, oz, inv; int dx, dy, dz, ncell; };

__device__ __forceinline__ int gg_bg_axis(float x, float o, float inv, int d)
{
    // monotone in x; NaN -> 0; huge values clamp to the box
    const float f = floorf((x - o) * inv);
    return (int)fminf(fmaxf(f, 0.f), (float)(d - 1));
}

__global__ __launch_bounds__(1024) void gg_k_ball_grid_build(const float *__restrict__ known,
                                                             const int *__restrict__ downnum,
                                                             int m, int sk, float radius,
                                                             GGBallGridInfo *__restrict__ info,
                                                             int *__restrict__ cellStart,
                                                             float4 *__restrict__ sorted)
{
    extern __shared__ int cnt[];                 // [ncell] counts -> cursors
    __shared__ float rmin[3][16], rmax[3][16];
    __shared__ int swc[16];
    __shared__ GGBallGridInfo gi;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int dn = downnum[b];
    dn = dn > m ? m : (dn < 0 ? 0 : dn);
    const float *kb = known + (size_t)b * m * sk;   // sk floats per point (3, or 4: rows x y z w)
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int j = tid; j < dn; j += 1024) {
        const float x = kb[(size_t)j * sk], y = kb[(size_t)j * sk + 1], z = kb[(size_t)j * sk + 2];
        if (isfinite(x) && isfinite(y) && isfinite(z)) {
            mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);
            mn[1] = fminf(mn[1], y); mx[1] = fmaxf(mx[1], y);
            mn[2] = fminf(mn[2], z); mx[2] = fmaxf(mx[2], z);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64));
        }
        if (lane == 0) { rmin[a][wave] = mn[a]; rmax[a][wave] = mx[a]; }
    }
    __syncthreads();
    if (tid == 0) {
        float lo[3], hi[3];
        for (int a = 0; a < 3; a++) {
            lo[a] = FLT_MAX; hi[a] = -FLT_MAX;
            for (int w = 0; w < 16; w++) { lo[a] = fminf(lo[a], rmin[a][w]); hi[a] = fmaxf(hi[a], rmax[a][w]); }
        }
        GGBallGridInfo g;
        if (!(hi[0] >= lo[0])) {                 // no finite point
            g.ox = g.oy = g.oz = 0.f; g.inv = 0.f; g.dx = g.dy = g.dz = 1;
        } else {
            const float ext = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
            float cell = fmaxf(radius * 1.001f, ext / (float)GG_BG_DMAX);
            if (!(cell > 0.f) || !isfinite(cell)) cell = 1.f;
            g.inv = 1.f / cell;
            g.ox = lo[0]; g.oy = lo[1]; g.oz = lo[2];
            g.dx = (int)fminf(floorf((hi[0] - lo[0]) * g.inv) + 1.f, (float)GG_BG_DMAX);
            g.dy = (int)fminf(floorf((hi[1] - lo[1]) * g.inv) + 1.f, (float)GG_BG_DMAX);
            g.dz = (int)fminf(floorf((hi[2] - lo[2]) * g.inv) + 1.f, (float)GG_BG_DMAX);
        }
        g.ncell = g.dx * g.dy * g.dz;
        gi = g;
        info[b] = g;
    }
    __syncthreads();
    const GGBallGridInfo g = gi;
    for (int c = tid; c < g.ncell; c += 1024) cnt[c] = 0;
    __syncthreads();
    for (int j = tid; j < dn; j += 1024) {
        const float x = kb[(size_t)j * sk], y = kb[(size_t)j * sk + 1], z = kb[(size_t)j * sk + 2];
        if (isfinite(x) && isfinite(y) && isfinite(z)) {
            const int c = (gg_bg_axis(z, g.oz, g.inv, g.dz) * g.dy + gg_bg_axis(y, g.oy, g.inv, g.dy)) * g.dx +
                          gg_bg_axis(x, g.ox, g.inv, g.dx);
            atomicAdd(&cnt[c], 1);
        }
    }
    __syncthreads();
    // exclusive scan of cnt[0..ncell): contiguous runs per thread
    const int per = (g.ncell + 1023) / 1024;
    const int j0 = tid * per;
    int s = 0;
    for (int j = j0; j < j0 + per && j < g.ncell; j++) s += cnt[j];
    int incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) swc[wave] = incl;
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < wave; w++) wbase += swc[w];
    int run = wbase + incl - s;
    int *cs = cellStart + (size_t)b * (GG_BG_NCMAX + 1);
    for (int j = j0; j < j0 + per && j < g.ncell; j++) {
        const int c = cnt[j];
        cs[j] = run;
        cnt[j] = run;                            // becomes the scatter cursor
        run += c;
    }
    if (tid == 1023) cs[g.ncell] = run;
    __syncthreads();
    // the sorted list carries the coordinates with the index: the query reads ONE contiguous run of
    // 16-byte records per (z, y) row of cells instead of an index and then three scattered floats
    float4 *sb = sorted + (size_t)b * m;
    for (int j = tid; j < dn; j += 1024) {
        const float x = kb[(size_t)j * sk], y = kb[(size_t)j * sk + 1], z = kb[(size_t)j * sk + 2];
        if (isfinite(x) && isfinite(y) && isfinite(z)) {
            const int c = (gg_bg_axis(z, g.oz, g.inv, g.dz) * g.dy + gg_bg_axis(y, g.oy, g.inv, g.dy)) * g.dx +
                          gg_bg_axis(x, g.ox, g.inv, g.dx);
            sb[atomicAdd(&cnt[c], 1)] = make_float4(x, y, z, __int_as_float(j));
        }
    }
}

template <int K>
__global__ __launch_bounds__(256) void gg_k_ball_grid_query(const float *__restrict__ unknown,
                                                            const int *__restrict__ upnum, int n, int su, int ztail,
                                                            int m, int topk, float r2,
                                                            const GGBallGridInfo *__restrict__ info,
                                                            const int *__restrict__ cellStart,
                                                            const float4 *__restrict__ sorted,
                                                            int *__restrict__ idx)
{
    const int b = blockIdx.y;
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= n) return;
    if (qi >= upnum[b]) {                        // rows >= upnum are not written (as the reference) ...
        if (ztail)                               // ... or zeroed, for callers that hand over raw memory
            for (int l = 0; l < topk; l++) idx[((size_t)b * n + qi) * topk + l] = 0;
        return;
    }
    const GGBallGridInfo g = info[b];
    const float *u = unknown + ((size_t)b * n + qi) * su;
    const float ux = u[0], uy = u[1], uz = u[2];
    const int *cs = cellStart + (size_t)b * (GG_BG_NCMAX + 1);
    const float4 *sb = sorted + (size_t)b * m;
    float best[K];
    int besti[K];
#pragma unroll
    for (int l = 0; l < K; l++) { best[l] = FLT_MAX; besti[l] = -1; }
    auto candidate = [&](const float4 kp) {
        const int id = __float_as_int(kp.w);
        const float dx = __fsub_rn(ux, kp.x);
        const float dy = __fsub_rn(uy, kp.y);
        const float dz = __fsub_rn(uz, kp.z);
        const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)),
                                  __fmul_rn(dz, dz));
        if (d > r2) return;                                        // ball_k_nn-inl.h:77
        // keep the K smallest by (d, id): what the reference's strict-< insertion over
        // ascending ids produces
        bool lt[K];
#pragma unroll
        for (int l = 0; l < K; l++)
            lt[l] = d < best[l] || (d == best[l] && (unsigned)id < (unsigned)besti[l]);
        if (lt[K - 1]) {
#pragma unroll
            for (int l = K - 1; l >= 1; l--) {
                besti[l] = lt[l - 1] ? besti[l - 1] : (lt[l] ? id : besti[l]);
                best[l] = lt[l - 1] ? best[l - 1] : (lt[l] ? d : best[l]);
            }
            besti[0] = lt[0] ? id : besti[0];
            best[0] = lt[0] ? d : best[0];
        }
    };
    const int cx = gg_bg_axis(ux, g.ox, g.inv, g.dx), cy = gg_bg_axis(uy, g.oy, g.inv, g.dy);
    const int cz = gg_bg_axis(uz, g.oz, g.inv, g.dz);
    // the x-neighbours are contiguous cells: one range of records per (z, y) row of cells; the nine
    // pairs of range bounds are requested together, before the first of them is needed
    const int x0 = cx > 0 ? cx - 1 : 0, x1 = cx + 1 < g.dx ? cx + 1 : g.dx - 1;
    int pa[9], pb[9];
#pragma unroll
    for (int r = 0; r < 9; r++) {
        const int z = cz + r / 3 - 1, y = cy + r % 3 - 1;
        const bool in = (z >= 0) & (z < g.dz) & (y >= 0) & (y < g.dy);
        const int c0 = ((in ? z : cz) * g.dy + (in ? y : cy)) * g.dx;
        const int a = cs[c0 + x0], e = cs[c0 + x1 + 1];
        pa[r] = in ? a : 0;
        pb[r] = in ? e : 0;
    }
#pragma unroll
    for (int r = 0; r < 9; r++) {
        const int p0 = pa[r], p1 = pb[r];
        // four records in flight (the one-at-a-time walk was a chain of L2 latencies)
        for (int q = p0; q < p1; q += 4) {
            float4 kp[4];
#pragma unroll
            for (int u = 0; u < 4; u++) kp[u] = sb[q + u < p1 ? q + u : p1 - 1];
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (q + u < p1) candidate(kp[u]);
        }
    }
    int *o = idx + ((size_t)b * n + qi) * topk;
#pragma unroll
    for (int l = 0; l < K; l++)
        if (l < topk) o[l] = besti[l];
}

// Few queries (the coarse up layers: 2 K - 8 K unknown points): one thread per query is one serial chain
// of ~50 top-k insertions on a handful of workgroups (59 us for 8192 queries at cfg4 up1 -- as long as the
// 655 360 queries of up2).  Here L consecutive lanes share a query: candidate f of the query's flat
// candidate list (its nine record runs one after the other) goes to lane f mod L, every lane keeps its own
// top-K, and the L lists are merged by K rounds of a group-wide minimum over (distance, index) -- the same
// total order as above, so the result is the same K records whatever L is.
template <int K, int L>
__global__ __launch_bounds__(256) void gg_k_ball_grid_query_ml(const float *__restrict__ unknown,
                                                               const int *__restrict__ upnum, int n, int su, int ztail,
                                                               int m, int topk, float r2,
                                                               const GGBallGridInfo *__restrict__ info,
                                                               const int *__restrict__ cellStart,
                                                               const float4 *__restrict__ sorted,
                                                               int *__restrict__ idx)
{
    const int b = blockIdx.y;
    const int qi = (int)((blockIdx.x * 256 + threadIdx.x) / L), sub = (int)(threadIdx.x % L);
    const bool act = qi < n && qi < upnum[b];    // (the same for the L lanes of a group; no early return:
                                                 //  the merge shuffles need every lane)
    const int qc = qi < n ? qi : n - 1;
    const GGBallGridInfo g = info[b];
    const float *u = unknown + ((size_t)b * n + qc) * su;
    const float ux = u[0], uy = u[1], uz = u[2];
    const int *cs = cellStart + (size_t)b * (GG_BG_NCMAX + 1);
    const float4 *sb = sorted + (size_t)b * m;
    float best[K];
    int besti[K];
#pragma unroll
    for (int l = 0; l < K; l++) { best[l] = FLT_MAX; besti[l] = -1; }
    const int cx = gg_bg_axis(ux, g.ox, g.inv, g.dx), cy = gg_bg_axis(uy, g.oy, g.inv, g.dy);
    const int cz = gg_bg_axis(uz, g.oz, g.inv, g.dz);
    const int x0 = cx > 0 ? cx - 1 : 0, x1 = cx + 1 < g.dx ? cx + 1 : g.dx - 1;
    int pa[9], pb[9];
#pragma unroll
    for (int r = 0; r < 9; r++) {
        const int z = cz + r / 3 - 1, y = cy + r % 3 - 1;
        const bool in = act & (z >= 0) & (z < g.dz) & (y >= 0) & (y < g.dy);
        const int c0 = ((in ? z : cz) * g.dy + (in ? y : cy)) * g.dx;
        const int a = cs[c0 + x0], e = cs[c0 + x1 + 1];
        pa[r] = in ? a : 0;
        pb[r] = in ? e : 0;
    }
    int off = 0;                                  // flat position of the run's first record
#pragma unroll
    for (int r = 0; r < 9; r++) {
        const int p0 = pa[r], p1 = pb[r];
        for (int q = p0 + ((sub - off) & (L - 1)); q < p1; q += L) {
            const float4 kp = sb[q];
            const int id = __float_as_int(kp.w);
            const float dx = __fsub_rn(ux, kp.x);
            const float dy = __fsub_rn(uy, kp.y);
            const float dz = __fsub_rn(uz, kp.z);
            const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            if (d > r2) continue;                                  // ball_k_nn-inl.h:77
            bool lt[K];
#pragma unroll
            for (int l = 0; l < K; l++)
                lt[l] = d < best[l] || (d == best[l] && (unsigned)id < (unsigned)besti[l]);
            if (lt[K - 1]) {
#pragma unroll
                for (int l = K - 1; l >= 1; l--) {
                    besti[l] = lt[l - 1] ? besti[l - 1] : (lt[l] ? id : besti[l]);
                    best[l] = lt[l - 1] ? best[l - 1] : (lt[l] ? d : best[l]);
                }
                besti[0] = lt[0] ? id : besti[0];
                best[0] = lt[0] ? d : best[0];
            }
        }
        off += p1 - p0;
    }
    // merge: K rounds, the group's smallest head by (d, id) leaves its list (empty = (FLT_MAX, -1) sorts last)
    int res[K];
#pragma unroll
    for (int l = 0; l < K; l++) {
        float d = best[0];
        int id = besti[0];
#pragma unroll
        for (int o = L / 2; o >= 1; o >>= 1) {
            const float od = __shfl_xor(d, o, 64);
            const int oi = __shfl_xor(id, o, 64);
            const bool take = od < d || (od == d && (unsigned)oi < (unsigned)id);
            d = take ? od : d;
            id = take ? oi : id;
        }
        res[l] = id;
        if (id != -1 && besti[0] == id) {
#pragma unroll
            for (int j = 0; j + 1 < K; j++) { best[j] = best[j + 1]; besti[j] = besti[j + 1]; }
            best[K - 1] = FLT_MAX;
            besti[K - 1] = -1;
        }
    }
    if (act && sub == 0) {
        int *o = idx + ((size_t)b * n + qi) * topk;
#pragma unroll
        for (int l = 0; l < K; l++)
            if (l < topk) o[l] = res[l];
    } else if (ztail && sub == 0 && qi < n) {
        for (int l = 0; l < topk; l++) idx[((size_t)b * n + qi) * topk + l] = 0;
    }
}

// workspace: info[B] | cellStart[B][NCMAX+1] | (16-byte aligned) sorted[B][m] (x, y, z, index)
static size_t gg_bg_sorted_offset(int B)
{
    const size_t o = (size_t)B * sizeof(GGBallGridInfo) + (size_t)B * (GG_BG_NCMAX + 1) * sizeof(int);
    return (o + 15) & ~(size_t)15;
}

size_t gg_ball_grid_workspace(int B, int m)
{
    return gg_bg_sorted_offset(B) + (size_t)B * m * sizeof(float4);
}

// 1 = not supported (k > 6): the caller uses the tiled scan
int gg_ball_knn_grid(const float *unknown, const float *known, const int *downnum,
                     const int *upnum, int B, int n, int m, int k, float radius, int *idx,
                     void *workspace, hipStream_t st, int su, int sk, int ztail)
{
    if (k < 1 || k > 6 || !(radius >= 0.f)) return 1;
    if ((uintptr_t)workspace & 15) return 1;      // float4 records at a 16-byte-aligned OFFSET from it
    GGBallGridInfo *info = (GGBallGridInfo *)workspace;
    int *cellStart = (int *)(info + B);
    float4 *sorted = (float4 *)((char *)workspace + gg_bg_sorted_offset(B));
    gg_k_ball_grid_build<<<B, 1024, GG_BG_NCMAX * sizeof(int), st>>>(known, downnum, m, sk, radius, info,
                                                                     cellStart, sorted);
    dim3 grid((n + 255) / 256, B);
    const float r2 = radius * radius;
    if ((long long)B * n <= 32768) {
        // few queries: 8 lanes per query (see gg_k_ball_grid_query_ml)
        dim3 g8(((unsigned)n * 8 + 255) / 256, B);
        if (k <= 3)
            gg_k_ball_grid_query_ml<3, 8><<<g8, 256, 0, st>>>(unknown, upnum, n, su, ztail, m, k, r2, info, cellStart, sorted, idx);
        else
            gg_k_ball_grid_query_ml<6, 8><<<g8, 256, 0, st>>>(unknown, upnum, n, su, ztail, m, k, r2, info, cellStart, sorted, idx);
        return hipGetLastError() == hipSuccess ? 0 : 3;
    }
    if (k <= 3)
        gg_k_ball_grid_query<3><<<grid, 256, 0, st>>>(unknown, upnum, n, su, ztail, m, k, r2, info,
                                                      cellStart, sorted, idx);
    else
        gg_k_ball_grid_query<6><<<grid, 256, 0, st>>>(unknown, upnum, n, su, ztail, m, k, r2, info,
                                                      cellStart, sorted, idx);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
