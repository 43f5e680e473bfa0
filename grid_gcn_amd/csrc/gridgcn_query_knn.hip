// gridgcn_query_knn.hip -- GridifyKNN neighbourhood query (gfx950, one wave64 per centre).
//
// Replaces gridifyKNN_kernel_query_neighs (gridifyop/gridifyknn.cu:206-333): exact top-P by
// squared distance to the voxel centre over Chebyshev shells, stable in the reference's
// traversal order (layer; w,h,d; bucket slot).  The reference keeps a per-thread 128-entry
// insertion-sorted array in local memory; here the candidates of the visited shells are staged
// in LDS and every candidate computes its final position directly:
//   position = #{candidates with smaller (distance, traversal number)}
// which equals the result of the reference's stable insertion sort.
#include "gridgcn_index.h"
#include <float.h>

#define GG_KNN_CAP 2048  // candidates staged per LDS tile

struct GGKnnPtrs {
    const int2 *vtab;
    const int *sorted, *bkt, *slotfirst1, *centnum, *exact;
};

__global__ __launch_bounds__(64) void gg_k_query_knn(const float4 *__restrict__ data, int N,
                                                     GGGrid gp, GGKnnPtrs q,
                                                     int *__restrict__ nebidx,
                                                     float *__restrict__ nebmsk,
                                                     float4 *__restrict__ cent,
                                                     float *__restrict__ centmsk)
{
    __shared__ int s_excl[GG_K3MAX + 1];
    __shared__ int s_off[GG_K3MAX];
    __shared__ int s_a[GG_K3MAX];
    __shared__ float s_d[GG_KNN_CAP];
    __shared__ int s_out[GG_PMAX];
    __shared__ float s_w[GG_PMAX];

    const int lane = threadIdx.x;
    const int index = blockIdx.x;
    const int b = index / gp.O;
    const int o = index - b * gp.O;
    const int P = gp.P, k = gp.k, k3 = gp.k3;
    const int cn = q.centnum[b];
    int *row = nebidx + (size_t)index * P;
    float *mrow = nebmsk + (size_t)index * P;
    if (o >= cn) {
        for (int s = lane; s < P; s += 64) { row[s] = 0; mrow[s] = 0.0f; }
        if (lane == 0) { cent[index] = make_float4(1.f, 1.f, 1.f, 1.f); centmsk[index] = 0.0f; }
        return;
    }
    const float4 *cloud = data + (size_t)b * N;
    const int i0 = q.slotfirst1[index] - 1;
    int c3[3];
    const float4 p0 = cloud[i0];
    const int v = gg_voxel_of(p0.x, p0.y, p0.z, gp, c3);
    const int c0 = c3[0], c1 = c3[1], c2 = c3[2];
    const int hk = (k - 1) / 2;
    // gridifyknn.cu:253-255: (int + 0.5) * voxel_size in double, coord_shift NOT subtracted
    const float ux = (float)((c0 + 0.5) * (double)gp.vs[0]);
    const float uy = (float)((c1 + 0.5) * (double)gp.vs[1]);
    const float uz = (float)((c2 + 0.5) * (double)gp.vs[2]);

    // ---- neighbour table in traversal order: layer-major, then (w,h,d) (:264-269) ----
    int nE = 0;                 // table entries so far
    int layer_end[GG_KMAX / 2 + 1];
    for (int L = 0; L <= hk; L++) {
        for (int base = 0; base < k3; base += 64) {
            int cell = base + lane;
            bool mine = false;
            int a = 0, so = 0;
            if (cell < k3) {
                int w = cell / (k * k) - hk;
                int h = (cell % (k * k)) / k - hk;
                int d = cell % k - hk;
                int aw = w < 0 ? -w : w, ah = h < 0 ? -h : h, ad = d < 0 ? -d : d;
                int mx = aw > ah ? aw : ah; mx = mx > ad ? mx : ad;
                mine = (mx == L);
                int dc = d + c2, hc = h + c1, wc = w + c0;
                if (mine && dc >= 0 && dc < gp.g[2] && hc >= 0 && hc < gp.g[1] && wc >= 0 &&
                    wc < gp.g[0]) {
                    size_t nb = (size_t)b * gp.G + (size_t)dc * gp.gxy + hc * gp.g[0] + wc;
                    const int2 vt = q.vtab[nb];
                    int c = vt.y;
                    a = c < P ? c : P;
                    so = vt.x | (c > P ? 0x80000000 : 0);
                }
            }
            unsigned long long m = __ballot(mine);
            int pos = nE + __popcll(m & ((1ull << lane) - 1ull));
            if (mine) { s_a[pos] = a; s_off[pos] = so; }
            nE += __popcll(m);
        }
        layer_end[L] = nE;
    }
    __syncthreads();
    int M = 0;
    for (int base = 0; base < nE; base += 64) {
        int e = base + lane;
        int a = e < nE ? s_a[e] : 0;
        int incl = gg_wave_incl_scan(a);
        if (e < nE) s_excl[e] = M + incl - a;
        M += __shfl(incl, 63, 64);
    }
    if (lane == 0) s_excl[nE] = M;
    for (int s = lane; s < P; s += 64) s_out[s] = -1;
    __syncthreads();
    // shells are visited until the running amount reaches P (:304-305)
    int Lstop = hk;
    for (int L = hk; L >= 0; L--)
        if (s_excl[layer_end[L]] >= P) Lstop = L;
    const int nEs = layer_end[Lstop];
    const int C = s_excl[nEs];
    const int ntile = (C + GG_KNN_CAP - 1) / GG_KNN_CAP;

    auto cand = [&](int g0, int &id) -> float {
        int lo = 0, hi = nEs;
        while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (s_excl[mid] <= g0) lo = mid; else hi = mid;
        }
        int so = s_off[lo];
        const int *src = (so < 0) ? q.bkt : q.sorted;
        id = src[(so & 0x7fffffff) + (g0 - s_excl[lo])];
        float4 p = cloud[id];
        float dx = __fsub_rn(ux, p.x), dy = __fsub_rn(uy, p.y), dz = __fsub_rn(uz, p.z);
        float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        return (d < FLT_MAX) ? d : __builtin_inff();  // never inserted by "dst < best[l]"
    };

    for (int qc = 0; qc < C; qc += 64) {
        const int g0 = qc + lane;
        int myid = -1;
        float myd = __builtin_inff();
        if (g0 < C) myd = cand(g0, myid);
        int r = 0;
        for (int t = 0; t < ntile; t++) {
            const int tb = t * GG_KNN_CAP;
            const int tn = C - tb < GG_KNN_CAP ? C - tb : GG_KNN_CAP;
            if (ntile > 1 || qc == 0) {
                __syncthreads();
                for (int j = lane; j < tn; j += 64) { int tmp; s_d[j] = cand(tb + j, tmp); }
                __syncthreads();
            }
            for (int j = 0; j < tn; j++) {
                float dj = s_d[j];
                r += (dj < myd) || (dj == myd && (tb + j) < g0);
            }
        }
        if (g0 < C && myd < FLT_MAX && r < P) s_out[r] = myid;
    }
    __syncthreads();
    // defined behaviour for slots the reference leaves uninitialised: besti[0]
    const int first = s_out[0];
    for (int s = lane; s < P; s += 64) {
        int id = s_out[s];
        if (id < 0) id = first;
        row[s] = id;
        mrow[s] = 1.0f;                       // :312 mask is 1 for all P slots
        s_w[s] = cloud[id].w;
    }
    __syncthreads();
    const bool exact = q.exact[b] != 0;
    float total;
    if (exact) {
        long long acc = 0;
        for (int s = lane; s < P; s += 64) acc += (long long)(int)s_w[s];
        total = (float)gg_wave_sum_ll(acc);
    } else {
        total = 0.0f;
        for (int s = 0; s < P; s++) total = __fadd_rn(total, (float)(int)s_w[s]);
    }
    float cx = 1.0f, cy = 1.0f, cz = 1.0f;
    if (gp.loc == 1) {
        size_t vb = (size_t)b * gp.G + v;
        const int2 vt = q.vtab[vb];
        int c = vt.y;
        int so = vt.x;
        float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
        for (int base = 0; base < c; base += 64) {
            int j = base + lane;
            float px = 0.f, py = 0.f, pz = 0.f, pw = 0.f;
            if (j < c) {
                float4 p = cloud[q.sorted[so + j]];
                px = __fmul_rn(p.x, p.w); py = __fmul_rn(p.y, p.w); pz = __fmul_rn(p.z, p.w);
                pw = p.w;
            }
            int nn = c - base < 64 ? c - base : 64;
            for (int l = 0; l < nn; l++) {
                sx = __fadd_rn(sx, __shfl(px, l, 64));
                sy = __fadd_rn(sy, __shfl(py, l, 64));
                sz = __fadd_rn(sz, __shfl(pz, l, 64));
                sw = __fadd_rn(sw, __shfl(pw, l, 64));
            }
        }
        cx = __fdiv_rn(sx, sw); cy = __fdiv_rn(sy, sw); cz = __fdiv_rn(sz, sw);
    }
    if (lane == 0) {
        cent[index] = make_float4(cx, cy, cz, total);
        centmsk[index] = 1.0f;
    }
}

int gg_launch_query_knn(const float *data, int B, int N, const GGGrid &gp, char *wsbase,
                        const GGIndexWs &w, int *nebidx, float *nebmsk, float *cent,
                        float *centmsk, const int *centnum, hipStream_t st)
{
    GGKnnPtrs q;
    q.vtab = (const int2 *)(wsbase + w.o_vtab);
    q.sorted = (const int *)(wsbase + w.o_sorted);
    q.bkt = (const int *)(wsbase + w.o_bkt);
    q.slotfirst1 = (const int *)(wsbase + w.o_slotfirst1);
    q.centnum = centnum;
    q.exact = (const int *)(wsbase + w.o_exact);
    gg_k_query_knn<<<B * gp.O, 64, 0, st>>>((const float4 *)data, N, gp, q, nebidx, nebmsk,
                                            (float4 *)cent, centmsk);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
