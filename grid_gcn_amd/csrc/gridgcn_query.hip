// gridgcn_query.hip -- neighbourhood query kernels (gfx950, one wave64 per centre / up point).
//
// Replaces gridify_kernel_query_neighs (gridifyop/gridify.cu:193-291) and the GridifyUp query
// (gridify_up.cu:172-225 together with its k^3 scatter build, :121-169).  The reference runs one
// THREAD per centre through a serial loop of up to k^3*P iterations with a curand_init per
// overflow item.  Here a wave owns the centre: lanes build the k^3 neighbour table (wave prefix
// sum of the bucket sizes), items are addressed flat (binary search in LDS), overflow items only
// evaluate their reservoir draw and the S0 outcome "the last writer of a slot wins" is an LDS
// atomicMax on the item number; outputs leave as coalesced rows.
#include "gridgcn_index.h"

struct GGQueryPtrs {
    const int *cnt, *off, *vox, *sorted, *bkt, *slotfirst1, *centnum, *exact;
};

// item g0 (0-based, flat over the neighbour table) -> point id
__device__ __forceinline__ int gg_item(const int *s_excl, const int *s_off, int k3, int g0,
                                       const int *__restrict__ sorted,
                                       const int *__restrict__ bkt)
{
    int lo = 0, hi = k3;  // s_excl[lo] <= g0 < s_excl[hi] (s_excl[k3] = total)
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (s_excl[mid] <= g0) lo = mid; else hi = mid;
    }
    int so = s_off[lo];
    int j = g0 - s_excl[lo];
    const int *src = (so < 0) ? bkt : sorted;
    return src[(so & 0x7fffffff) + j];
}

// grid = B*O blocks of 64 threads.
__global__ __launch_bounds__(64) void gg_k_query_gridify(const float4 *__restrict__ data, int N,
                                                         GGGrid gp, GGQueryPtrs q,
                                                         int *__restrict__ nebidx,
                                                         float *__restrict__ nebmsk,
                                                         float4 *__restrict__ cent,
                                                         float *__restrict__ centmsk)
{
    __shared__ int s_excl[GG_K3MAX + 1];
    __shared__ int s_off[GG_K3MAX];
    __shared__ int s_slotg[GG_PMAX];
    __shared__ int s_slotid[GG_PMAX];
    __shared__ float s_curw[GG_PMAX];

    const int lane = threadIdx.x;
    const int index = blockIdx.x;  // b*O + o
    const int b = index / gp.O;
    const int o = index - b * gp.O;
    const int P = gp.P, k = gp.k, k3 = gp.k3;
    const int cn = q.centnum[b];
    int *row = nebidx + (size_t)index * P;
    float *mrow = nebmsk + (size_t)index * P;

    if (o >= cn) {  // GridifyOp::Forward fill values (gridify-inl.h:117-121)
        for (int s = lane; s < P; s += 64) { row[s] = 0; mrow[s] = 0.0f; }
        if (lane == 0) { cent[index] = make_float4(1.f, 1.f, 1.f, 1.f); centmsk[index] = 0.0f; }
        return;
    }
    const float4 *cloud = data + (size_t)b * N;
    const int i0 = q.slotfirst1[index] - 1;
    const int v = q.vox[(size_t)b * N + i0];
    const int c2 = v / gp.gxy;                         // gridify.cu:232-234
    const int c1 = (v - c2 * gp.gxy) / gp.g[0];
    const int c0 = v - c2 * gp.gxy - c1 * gp.g[0];
    const int hk = (k - 1) / 2;

    // ---- neighbour table: flat item offsets over the k^3 voxels in (z,y,x) order (:240-249) ----
    int M = 0;
    for (int base = 0; base < k3; base += 64) {
        int nei = base + lane;
        int a = 0, so = 0;
        if (nei < k3) {
            int d = nei / (k * k) - hk + c2;
            int h = (nei % (k * k)) / k - hk + c1;
            int w = nei % k - hk + c0;
            if (d >= 0 && d < gp.g[2] && h >= 0 && h < gp.g[1] && w >= 0 && w < gp.g[0]) {
                size_t nb = (size_t)b * gp.G + (size_t)d * gp.gxy + h * gp.g[0] + w;
                int c = q.cnt[nb];
                a = c < P ? c : P;
                so = q.off[nb] | (c > P ? 0x80000000 : 0);
            }
        }
        int incl = gg_wave_incl_scan(a);
        if (nei < k3) { s_excl[nei] = M + incl - a; s_off[nei] = so; }
        M += __shfl(incl, 63, 64);
    }
    if (lane == 0) s_excl[k3] = M;
    for (int s = lane; s < P; s += 64) s_slotg[s] = 0;
    __syncthreads();

    const int Mc = M < P ? M : P;
    // ---- first P items fill slots 0..P-1 in order (:251-258) ----
    for (int g0 = lane; g0 < Mc; g0 += 64) {
        int id = gg_item(s_excl, s_off, k3, g0, q.sorted, q.bkt);
        s_slotid[g0] = id;
        s_curw[g0] = cloud[id].w;
    }
    __syncthreads();

    const bool exact = q.exact[b] != 0;
    const unsigned seedbase = (unsigned)index * (unsigned)P * (unsigned)k3;  // int wrap (:260)
    float total;
    if (exact) {
        // ---- overflow items: slot r(g) <- item g, last writer (largest g) wins (:260-268) ----
        for (int g0 = P + lane; g0 < M; g0 += 64) {
            int g = g0 + 1;
            int s32 = (int)(seedbase + (unsigned)g);
            int r = gg_reservoir_pick((unsigned long long)(long long)s32, g);
            if (r < P) atomicMax(&s_slotg[r], g);
        }
        __syncthreads();
        // integer weights: the running total of S0 telescopes to the sum over the final slots
        long long acc = 0;
        for (int s = lane; s < Mc; s += 64) {
            int g = s_slotg[s];
            int id = s_slotid[s];
            float w = s_curw[s];
            if (g > 0) {
                id = gg_item(s_excl, s_off, k3, g - 1, q.sorted, q.bkt);
                w = cloud[id].w;
                s_slotid[s] = id;
            }
            acc += (long long)(int)w;
        }
        acc = gg_wave_sum_ll(acc);
        total = (float)acc;
    } else {
        // ---- general weights: replay S0's float accumulation in its own order ----
        total = 0.0f;
        for (int s = 0; s < Mc; s++) total = __fadd_rn(total, (float)(int)s_curw[s]);
        for (int base = P; base < M; base += 64) {
            int g0 = base + lane;
            bool ev = false;
            int r = 0, idn = 0;
            float wn = 0.0f;
            if (g0 < M) {
                int g = g0 + 1;
                int s32 = (int)(seedbase + (unsigned)g);
                r = gg_reservoir_pick((unsigned long long)(long long)s32, g);
                ev = r < P;
            }
            if (ev) {
                idn = gg_item(s_excl, s_off, k3, g0, q.sorted, q.bkt);
                wn = cloud[idn].w;
            }
            unsigned long long mask = __ballot(ev);
            while (mask) {
                int l = __builtin_ctzll(mask);
                mask &= mask - 1;
                int rr = __shfl(r, l, 64);
                float wl = __shfl(wn, l, 64);
                int il = __shfl(idn, l, 64);
                float old = s_curw[rr];
                total = __fadd_rn(total, __fsub_rn((float)(int)wl, old));
                __syncthreads();
                if (lane == 0) { s_curw[rr] = wl; s_slotid[rr] = il; }
                __syncthreads();
            }
        }
    }
    __syncthreads();

    // ---- outputs: ids, mask, pad with the first id (:275-279) ----
    const int first = s_slotid[0];
    for (int s = lane; s < P; s += 64) {
        row[s] = s < Mc ? s_slotid[s] : first;
        mrow[s] = s < Mc ? 1.0f : 0.0f;
    }
    // ---- centre location: weighted mean of ALL points of the centre voxel, accumulated in
    //      ascending point id with separate multiply and add (:155-162, :280-289) ----
    float cx = 1.0f, cy = 1.0f, cz = 1.0f;
    if (gp.loc == 1) {
        size_t vb = (size_t)b * gp.G + v;
        int c = q.cnt[vb];
        int so = q.off[vb];
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

// ------------------------------------------------------------------------------------------
// GridifyUp query.  grid = B*O blocks of 64 threads (one wave per up point).
// S0 of the reference build: voxel nb receives every down point whose own voxel lies in the k^3
// neighbourhood of nb, in ascending point id; item n >= P overwrites slot
// r = pick(seed + threadindex, n+1) with threadindex = (b*Nd + id)*k^3 + nei (gridify_up.cu:121-167).
// That bucket is rebuilt here on the fly for the up point's voxel only, from the down points'
// own-voxel sorted segments: rank n of a candidate = number of candidates with a smaller id.
__global__ __launch_bounds__(64) void gg_k_query_up(const float4 *__restrict__ updata,
                                                    const int *__restrict__ up_np, int Nd,
                                                    GGGrid gp, GGQueryPtrs q,
                                                    int *__restrict__ nebidx,
                                                    float *__restrict__ nebmsk)
{
    __shared__ int s_excl[GG_K3MAX + 1];
    __shared__ int s_off[GG_K3MAX];
    __shared__ int s_cnt[GG_K3MAX];
    __shared__ int s_slot[GG_PMAX];

    const int lane = threadIdx.x;
    const int index = blockIdx.x;
    const int b = index / gp.O;
    const int o = index - b * gp.O;
    const int P = gp.P, k = gp.k, k3 = gp.k3;
    int *row = nebidx + (size_t)index * P;
    float *mrow = nebmsk + (size_t)index * P;

    int c3[3] = {0, 0, 0};
    int vq = -1;
    if (o < up_np[b]) {
        float4 p = updata[index];
        vq = gg_voxel_of(p.x, p.y, p.z, gp, c3);
    }
    if (vq < 0) {  // gridify_up-inl.h:111-112 fill values
        for (int s = lane; s < P; s += 64) { row[s] = 0; mrow[s] = 0.0f; }
        return;
    }
    const int hk = (k - 1) / 2;
    int M = 0;
    for (int base = 0; base < k3; base += 64) {
        int nei = base + lane;
        int a = 0, so = 0;
        if (nei < k3) {
            int d = nei / (k * k) - hk + c3[2];
            int h = (nei % (k * k)) / k - hk + c3[1];
            int w = nei % k - hk + c3[0];
            if (d >= 0 && d < gp.g[2] && h >= 0 && h < gp.g[1] && w >= 0 && w < gp.g[0]) {
                size_t nb = (size_t)b * gp.G + (size_t)d * gp.gxy + h * gp.g[0] + w;
                a = q.cnt[nb];
                so = q.off[nb];
            }
        }
        int incl = gg_wave_incl_scan(a);
        if (nei < k3) { s_excl[nei] = M + incl - a; s_off[nei] = so; s_cnt[nei] = a; }
        M += __shfl(incl, 63, 64);
    }
    if (lane == 0) s_excl[k3] = M;
    for (int s = lane; s < P; s += 64) s_slot[s] = -1;
    __syncthreads();

    for (int g0 = lane; g0 < M; g0 += 64) {
        // locate the item
        int lo = 0, hi = k3;
        while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (s_excl[mid] <= g0) lo = mid; else hi = mid;
        }
        int id = q.sorted[s_off[lo] + (g0 - s_excl[lo])];
        // rank among all candidates: sum over the (ascending) segments of #entries < id
        int n = 0;
        for (int e = 0; e < k3; e++) {
            int c = s_cnt[e];
            if (c == 0) continue;
            const int *sp = q.sorted + s_off[e];
            int l2 = 0, h2 = c;  // first position with sp[pos] >= id
            while (l2 < h2) {
                int mid = (l2 + h2) >> 1;
                if (sp[mid] < id) l2 = mid + 1; else h2 = mid;
            }
            n += l2;
        }
        int s = n;
        if (n >= P) {
            // the scatter thread of this (point, voxel) pair used offset index nei' with
            // own_voxel + offset(nei') = query voxel, i.e. the mirror of `lo`
            long long threadindex = ((long long)b * Nd + id) * k3 + (k3 - 1 - lo);
            s = gg_reservoir_pick(gp.seed + (unsigned long long)threadindex, n + 1);
        }
        if (s < P) atomicMax(&s_slot[s], id);
    }
    __syncthreads();
    const int first = M > 0 ? s_slot[0] : 0;  // reference: uninitialised initID when M == 0
    for (int s = lane; s < P; s += 64) {
        row[s] = s < M ? s_slot[s] : first;
        mrow[s] = s < M ? 1.0f : 0.0f;
    }
}

// ------------------------------------------------------------------------------------------
// host launchers (used by gridgcn_capi.hip)
int gg_launch_query_gridify(const float *data, int B, int N, const GGGrid &gp, char *wsbase,
                            const GGIndexWs &w, int *nebidx, float *nebmsk, float *cent,
                            float *centmsk, const int *centnum, hipStream_t st)
{
    GGQueryPtrs q;
    q.cnt = (const int *)(wsbase + w.o_cnt);
    q.off = (const int *)(wsbase + w.o_off);
    q.vox = (const int *)(wsbase + w.o_vox);
    q.sorted = (const int *)(wsbase + w.o_sorted);
    q.bkt = (const int *)(wsbase + w.o_bkt);
    q.slotfirst1 = (const int *)(wsbase + w.o_slotfirst1);
    q.centnum = centnum;
    q.exact = (const int *)(wsbase + w.o_exact);
    gg_k_query_gridify<<<B * gp.O, 64, 0, st>>>((const float4 *)data, N, gp, q, nebidx, nebmsk,
                                                (float4 *)cent, centmsk);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_launch_query_up(const float *updata, const int *up_np, int B, int Nd, const GGGrid &gp,
                       char *wsbase, const GGIndexWs &w, int *nebidx, float *nebmsk,
                       hipStream_t st)
{
    GGQueryPtrs q = {};
    q.cnt = (const int *)(wsbase + w.o_cnt);
    q.off = (const int *)(wsbase + w.o_off);
    q.vox = (const int *)(wsbase + w.o_vox);
    q.sorted = (const int *)(wsbase + w.o_sorted);
    gg_k_query_up<<<B * gp.O, 64, 0, st>>>((const float4 *)updata, up_np, Nd, gp, q, nebidx,
                                           nebmsk);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
