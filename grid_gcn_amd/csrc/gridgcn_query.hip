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
    const int2 *vtab;  // .x = segment start, .y = population
    const int *sorted, *bkt, *slotfirst1, *centnum, *exact;
};

#define GG_QW 4    // waves per workgroup of the Gridify query (they never synchronise)
#define GG_QCS 12  // ints of per-centre state in LDS

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

// inclusive max-scan across the 64 lanes of a wave
__device__ __forceinline__ int gg_wave_incl_max(int v)
{
    const int lane = gg_lane();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(v, d, 64);
        if (lane >= d) v = v > t ? v : t;
    }
    return v;
}

// Gridify query: a wave owns NC consecutive centre slots and walks them through the dependent
// levels of the lookup TOGETHER, so that NC times the loads are in flight per level:
//   A  slot -> first point of the chosen voxel -> voxel coordinates        (2 loads, lanes < NC)
//   B  (start, population) of the k^3 neighbour voxels of every centre      (lane = neighbour)
//   C  exclusive scans = flat item offsets (:240-249)
//   D  ids of the first min(M,P) items (:251-258) and of the centre voxel's own points
//   E  their weights (skipped when every weight of the cloud is 1) and the own-voxel points
//   F  reservoir past P (:260-268), total weight, padding (:275-279), centre (:280-289)
// grid = B * ceil(O / (4*NC)) workgroups of 4 independent waves; dynamic LDS = 4 * NC *
// (GG_QCS + 2*k^3 + 1 + 3*P + 64) ints.
// XCD-aware mapping: workgroup w runs on XCD w % 8 (observed dispatch order; only speed depends on
// it).  With B a multiple of 8 all workgroups of an XCD work on the clouds b = XCD (mod 8), so the
// voxel table, the sorted ids and the points a cloud's centres share stay in ONE 4 MB L2 instead
// of being fetched into all eight (cfg5: 92 MB fetched by this kernel with the linear mapping).
template <int NC>
__global__ __launch_bounds__(64 * GG_QW) void gg_k_query_gridify(
    const float4 *__restrict__ data, int N, GGGrid gp, GGQueryPtrs q, int B,
    int *__restrict__ nebidx, float *__restrict__ nebmsk, float4 *__restrict__ cent,
    float *__restrict__ centmsk)
{
    extern __shared__ __attribute__((aligned(16))) int qlds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int P = gp.P, k = gp.k, k3 = gp.k3, hk = (k - 1) / 2;
    int *wl = qlds + wave * (NC * (GG_QCS + 2 * k3 + 1 + 3 * P + 64));
    int *s_ctr = wl;                         // [NC][QCS]: c0 c1 c2 b state exact ccnt coff total
    int *s_excl = s_ctr + NC * GG_QCS;       // [NC][k3+1]
    int *s_off = s_excl + NC * (k3 + 1);     // [NC][k3]
    int *s_slotg = s_off + NC * k3;          // [NC][P]
    int *s_slotid = s_slotg + NC * P;        // [NC][P]
    float *s_curw = (float *)(s_slotid + NC * P);   // [NC][P]
    float *s_mem = s_curw + NC * P;          // [NC][16][4] w*x, w*y, w*z, w of the own-voxel points
    const int wgpc = (gp.O + GG_QW * NC - 1) / (GG_QW * NC);  // workgroups per cloud
    int wb, wg;
    if ((B & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        wb = (j / wgpc) * 8 + xcd;
        wg = j % wgpc;
    } else {
        wb = blockIdx.x / wgpc;
        wg = blockIdx.x % wgpc;
    }
    const int o0 = (wg * GG_QW + wave) * NC;       // first centre slot of this wave in cloud wb
    const int cbase = wb * gp.O + o0;
    GG_STAMP(3, blockIdx.x, 0);
    if (o0 >= gp.O) return;

    // ---- A ----
    if (lane < NC) {
        const int index = cbase + lane;
        int c3[3] = {0, 0, 0};
        int b = wb, state = 0, ex = 0;  // state 0: beyond the cloud's O slots, 1: empty slot (fill values), 2: centre
        if (o0 + lane < gp.O) {
            const int o = o0 + lane;
            const int cn = q.centnum[b];
            const int sf = q.slotfirst1[index];
            ex = q.exact[b];
            state = 1;
            if (o < cn) {
                const float4 p0 = data[(size_t)b * N + (sf - 1)];
                (void)gg_voxel_of(p0.x, p0.y, p0.z, gp, c3);  // gridify.cu:232-234
                state = 2;
            }
        }
        int *ct = s_ctr + lane * GG_QCS;
        ct[0] = c3[0]; ct[1] = c3[1]; ct[2] = c3[2]; ct[3] = b; ct[4] = state; ct[5] = ex;
        ct[6] = 0; ct[7] = 0;
    }
    for (int s = lane; s < NC * P; s += 64) s_slotg[s] = 0;
    __builtin_amdgcn_wave_barrier();
    GG_STAMP(3, blockIdx.x, 1);

    // ---- B + C: neighbour voxels in (z,y,x) order (:240-249) ----
    int Mr[NC];
    if (k3 <= 64) {
        // lane = neighbour number; the table entries of all NC centres are loaded together
        const int kk = k * k;
        const int dz = lane / kk - hk, dy = (lane % kk) / k - hk, dx = lane % k - hk;
        int2 vt[NC];
        bool inb[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const int *ct = s_ctr + c * GG_QCS;
            const int d = dz + ct[2], h = dy + ct[1], w = dx + ct[0];
            inb[c] = lane < k3 && ct[4] == 2 && d >= 0 && d < gp.g[2] && h >= 0 && h < gp.g[1] &&
                     w >= 0 && w < gp.g[0];
            vt[c] = make_int2(0, 0);
            if (inb[c])
                vt[c] = q.vtab[(size_t)ct[3] * gp.G + (size_t)d * gp.gxy + h * gp.g[0] + w];
        }
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const int cn = inb[c] ? vt[c].y : 0;
            const int a = cn < P ? cn : P;
            if (inb[c] && lane == (k3 - 1) / 2) {  // the centre voxel itself
                s_ctr[c * GG_QCS + 6] = cn;
                s_ctr[c * GG_QCS + 7] = vt[c].x;
            }
            const int incl = gg_wave_incl_scan(a);
            if (lane < k3) {
                s_excl[c * (k3 + 1) + lane] = incl - a;
                s_off[c * k3 + lane] = vt[c].x | (cn > P ? 0x80000000 : 0);
                // head marker of the neighbour's first item (consumed in D)
                if (a > 0 && incl - a < P) s_slotg[c * P + incl - a] = lane + 1;
            }
            Mr[c] = __shfl(incl, 63, 64);
            if (lane == 0) s_excl[c * (k3 + 1) + k3] = Mr[c];
        }
    } else {
        const int kk = k * k;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const int *ct = s_ctr + c * GG_QCS;
            int M = 0;
            for (int base = 0; base < k3; base += 64) {
                const int nei = base + lane;
                int a = 0, so = 0, cn = 0;
                if (nei < k3 && ct[4] == 2) {
                    const int d = nei / kk - hk + ct[2];
                    const int h = (nei % kk) / k - hk + ct[1];
                    const int w = nei % k - hk + ct[0];
                    if (d >= 0 && d < gp.g[2] && h >= 0 && h < gp.g[1] && w >= 0 && w < gp.g[0]) {
                        const int2 vt = q.vtab[(size_t)ct[3] * gp.G + (size_t)d * gp.gxy + h * gp.g[0] + w];
                        cn = vt.y;
                        a = cn < P ? cn : P;
                        so = vt.x | (cn > P ? 0x80000000 : 0);
                        if (nei == (k3 - 1) / 2) {
                            s_ctr[c * GG_QCS + 6] = cn;
                            s_ctr[c * GG_QCS + 7] = vt.x;
                        }
                    }
                }
                const int incl = gg_wave_incl_scan(a);
                if (nei < k3) {
                    s_excl[c * (k3 + 1) + nei] = M + incl - a;
                    s_off[c * k3 + nei] = so;
                    if (a > 0 && M + incl - a < P) s_slotg[c * P + M + incl - a] = nei + 1;
                }
                M += __shfl(incl, 63, 64);
            }
            if (lane == 0) s_excl[c * (k3 + 1) + k3] = M;
            Mr[c] = M;
        }
    }
    __builtin_amdgcn_wave_barrier();
    GG_STAMP(3, blockIdx.x, 2);

    // ---- D: item g0 belongs to the last head marker at or before it ----
    int idr[NC][2], cidr[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int Mc = Mr[c] < P ? Mr[c] : P;
        int carry = 0;
#pragma unroll
        for (int g = 0; g < 2; g++) {
            const int g0 = g * 64 + lane;
            idr[c][g] = -1;
            if (g * 64 < Mc) {  // uniform
                int hd = g0 < P ? s_slotg[c * P + g0] : 0;
                hd = gg_wave_incl_max(hd);
                hd = hd > carry ? hd : carry;
                carry = __shfl(hd, 63, 64);
                if (g0 < Mc) {
                    const int e = hd - 1;
                    const int so = s_off[c * k3 + e];
                    const int *src = (so < 0) ? q.bkt : q.sorted;
                    idr[c][g] = src[(so & 0x7fffffff) + g0 - s_excl[c * (k3 + 1) + e]];
                }
            }
        }
        cidr[c] = -1;
        const int *ct = s_ctr + c * GG_QCS;
        if (gp.loc == 1 && ct[4] == 2 && lane < ct[6] && lane < 16) cidr[c] = q.sorted[ct[7] + lane];
    }
    __builtin_amdgcn_wave_barrier();
    for (int s = lane; s < NC * P; s += 64) s_slotg[s] = 0;  // markers consumed
    // ---- E ----
    float wr[NC][2];
    float4 cp[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int *ct = s_ctr + c * GG_QCS;
        const float4 *cloud = data + (size_t)ct[3] * N;
        const bool allone = (ct[5] & 2) != 0;
#pragma unroll
        for (int g = 0; g < 2; g++) {
            wr[c][g] = 1.0f;
            if (idr[c][g] >= 0 && !allone) wr[c][g] = cloud[idr[c][g]].w;
        }
        cp[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cidr[c] >= 0) cp[c] = cloud[cidr[c]];
    }
#pragma unroll
    for (int c = 0; c < NC; c++) {
#pragma unroll
        for (int g = 0; g < 2; g++)
            if (idr[c][g] >= 0) {
                s_slotid[c * P + g * 64 + lane] = idr[c][g];
                s_curw[c * P + g * 64 + lane] = wr[c][g];
            }
        if (lane < 16) {
            float *m = s_mem + (c * 16 + lane) * 4;
            m[0] = __fmul_rn(cp[c].x, cp[c].w);
            m[1] = __fmul_rn(cp[c].y, cp[c].w);
            m[2] = __fmul_rn(cp[c].z, cp[c].w);
            m[3] = cp[c].w;
        }
    }
    __builtin_amdgcn_wave_barrier();
    GG_STAMP(3, blockIdx.x, 3);

    // ---- F: reservoir past P (:260-268) and total weight.  Clouds with small integer weights
    //      (always, in the reference models): the running total of S0 telescopes to the sum over
    //      the final slots, so the levels below run for all NC centres together ----
    // F1: overflow items only evaluate their draw; slot r(g) <- item g, largest g wins
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int *ct = s_ctr + c * GG_QCS;
        if (ct[4] == 2 && (ct[5] & 1) && Mr[c] > P) {
            const unsigned seedbase = (unsigned)(cbase + c) * (unsigned)P * (unsigned)k3;  // int wrap (:260)
            for (int g0 = P + lane; g0 < Mr[c]; g0 += 64) {
                const int g = g0 + 1;
                const int s32 = (int)(seedbase + (unsigned)g);
                const int r = gg_reservoir_pick((unsigned long long)(long long)s32, g);
                if (r < P) atomicMax(&s_slotg[c * P + r], g);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // F2: ids of the winners
    int rid[NC][2];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int *ct = s_ctr + c * GG_QCS;
        const bool ov = ct[4] == 2 && (ct[5] & 1) && Mr[c] > P;
#pragma unroll
        for (int g2 = 0; g2 < 2; g2++) {
            const int sl = g2 * 64 + lane;
            rid[c][g2] = -1;
            if (ov && sl < P) {
                const int g = s_slotg[c * P + sl];
                if (g > 0)
                    rid[c][g2] = gg_item(s_excl + c * (k3 + 1), s_off + c * k3, k3, g - 1, q.sorted, q.bkt);
            }
        }
    }
    // F3: their weights
    float rw[NC][2];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int *ct = s_ctr + c * GG_QCS;
        const float4 *cloud = data + (size_t)ct[3] * N;
#pragma unroll
        for (int g2 = 0; g2 < 2; g2++) {
            rw[c][g2] = 1.0f;
            if (rid[c][g2] >= 0 && !(ct[5] & 2)) rw[c][g2] = cloud[rid[c][g2]].w;
        }
    }
#pragma unroll
    for (int c = 0; c < NC; c++)
#pragma unroll
        for (int g2 = 0; g2 < 2; g2++)
            if (rid[c][g2] >= 0) {
                s_slotid[c * P + g2 * 64 + lane] = rid[c][g2];
                s_curw[c * P + g2 * 64 + lane] = rw[c][g2];
            }
    __builtin_amdgcn_wave_barrier();
    // F4: totals (and, for clouds with general weights, the replay of S0's float accumulation)
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int index = cbase + c;
        int *ct = s_ctr + c * GG_QCS;
        if (ct[4] != 2) continue;
        const bool exact = (ct[5] & 1) != 0, allone = (ct[5] & 2) != 0;
        const int M = Mr[c], Mc = M < P ? M : P;
        float *curw = s_curw + c * P;
        float total;
        if (exact) {
            if (allone) {
                total = (float)Mc;  // every weight is 1
            } else {
                long long acc = 0;
                for (int sl = lane; sl < Mc; sl += 64) acc += (long long)(int)curw[sl];
                total = (float)gg_wave_sum_ll(acc);
            }
        } else {
            const float4 *cloud = data + (size_t)ct[3] * N;
            const int *ex = s_excl + c * (k3 + 1), *sof = s_off + c * k3;
            int *slotid = s_slotid + c * P;
            const unsigned seedbase = (unsigned)index * (unsigned)P * (unsigned)k3;
            total = 0.0f;
            for (int sl = 0; sl < Mc; sl++) total = __fadd_rn(total, (float)(int)curw[sl]);
            for (int base = P; base < M; base += 64) {
                const int g0 = base + lane;
                bool ev = false;
                int r = 0, idn = 0;
                float wn = 0.0f;
                if (g0 < M) {
                    const int g = g0 + 1;
                    const int s32 = (int)(seedbase + (unsigned)g);
                    r = gg_reservoir_pick((unsigned long long)(long long)s32, g);
                    ev = r < P;
                }
                if (ev) {
                    idn = gg_item(ex, sof, k3, g0, q.sorted, q.bkt);
                    wn = cloud[idn].w;
                }
                unsigned long long mask = __ballot(ev);
                while (mask) {
                    const int l = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    const int rr = __shfl(r, l, 64);
                    const float wl_ = __shfl(wn, l, 64);
                    const int il = __shfl(idn, l, 64);
                    const float old = curw[rr];
                    total = __fadd_rn(total, __fsub_rn((float)(int)wl_, old));
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) { curw[rr] = wl_; slotid[rr] = il; }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (lane == 0) ct[8] = __float_as_int(total);
    }
    __builtin_amdgcn_wave_barrier();
    // ---- outputs: ids, mask, pad with the first id (:275-279); fill values of empty slots ----
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int index = cbase + c;
        const int state = s_ctr[c * GG_QCS + 4];
        if (state == 0) break;
        int *row = nebidx + (size_t)index * P;
        float *mrow = nebmsk + (size_t)index * P;
        const int Mc = state == 2 ? (Mr[c] < P ? Mr[c] : P) : 0;
        const int first = state == 2 ? s_slotid[c * P] : 0;  // GridifyOp::Forward fill: 0
        for (int s = lane; s < P; s += 64) {
            row[s] = s < Mc ? s_slotid[c * P + s] : first;
            mrow[s] = s < Mc ? 1.0f : 0.0f;
        }
    }
    // ---- centre location: weighted mean of ALL points of the centre voxel, accumulated in
    //      ascending point id with separate multiply and add (:155-162, :280-289); lane c sums
    //      the points of centre c (the common case of <= 16 points), all centres at once ----
    bool big = false;
    if (lane < NC && o0 + lane < gp.O) {
        const int index = cbase + lane;
        const int *ct = s_ctr + lane * GG_QCS;
        if (ct[4] == 2) {
            float cx = 1.0f, cy = 1.0f, cz = 1.0f;
            const int cc = ct[6];
            if (gp.loc == 1) {
                if (cc <= 16) {
                    float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
                    for (int l = 0; l < cc; l++) {
                        const float *m = s_mem + (lane * 16 + l) * 4;
                        sx = __fadd_rn(sx, m[0]); sy = __fadd_rn(sy, m[1]);
                        sz = __fadd_rn(sz, m[2]); sw = __fadd_rn(sw, m[3]);
                    }
                    cx = __fdiv_rn(sx, sw); cy = __fdiv_rn(sy, sw); cz = __fdiv_rn(sz, sw);
                } else {
                    big = true;
                }
            }
            if (!big) {
                cent[index] = make_float4(cx, cy, cz, __int_as_float(ct[8]));
                centmsk[index] = 1.0f;
            }
        } else {  // gridify-inl.h:117-121
            cent[index] = make_float4(1.f, 1.f, 1.f, 1.f);
            centmsk[index] = 0.0f;
        }
    }
    unsigned long long bigm = __ballot(big);
    while (bigm) {  // centre voxels with more than 16 points: the whole wave walks the segment
        const int c = __builtin_ctzll(bigm);
        bigm &= bigm - 1;
        const int *ct = s_ctr + c * GG_QCS;
        const float4 *cloud = data + (size_t)ct[3] * N;
        const int cc = ct[6], so = ct[7];
        float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
        for (int base = 0; base < cc; base += 64) {
            const int j = base + lane;
            float4 pp = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < cc) pp = cloud[q.sorted[so + j]];
            const float px = __fmul_rn(pp.x, pp.w), py = __fmul_rn(pp.y, pp.w),
                        pz = __fmul_rn(pp.z, pp.w), pw = pp.w;
            const int nn = cc - base < 64 ? cc - base : 64;
            for (int l = 0; l < nn; l++) {
                sx = __fadd_rn(sx, __shfl(px, l, 64));
                sy = __fadd_rn(sy, __shfl(py, l, 64));
                sz = __fadd_rn(sz, __shfl(pz, l, 64));
                sw = __fadd_rn(sw, __shfl(pw, l, 64));
            }
        }
        if (lane == 0) {
            cent[cbase + c] = make_float4(__fdiv_rn(sx, sw), __fdiv_rn(sy, sw), __fdiv_rn(sz, sw),
                                          __int_as_float(ct[8]));
            centmsk[cbase + c] = 1.0f;
        }
    }
    GG_STAMP(3, blockIdx.x, 4);
}
GG_PROF_SETTER(gridgcn_prof_set_query)

// ------------------------------------------------------------------------------------------
// GridifyUp query.  grid = B*O blocks of 64 threads (one wave per up point).
// S0 of the reference build: voxel nb receives every down point whose own voxel lies in the k^3
// neighbourhood of nb, in ascending point id; item n >= P overwrites slot
// r = pick(seed + threadindex, n+1) with threadindex = (b*Nd + id)*k^3 + nei (gridify_up.cu:121-167).
// That bucket is rebuilt here on the fly for the up point's voxel only, from the down points'
// own-voxel sorted segments: rank n of a candidate = number of candidates with a smaller id.
// one wave, one up point (general k, P): lanes enumerate the candidates
__device__ __forceinline__ void gg_query_up_wave(int index, const float4 *__restrict__ updata,
                                                 const int *__restrict__ up_np, int Nd,
                                                 const GGGrid &gp, const GGQueryPtrs &q,
                                                 int *__restrict__ nebidx, float *__restrict__ nebmsk,
                                                 int *s_excl, int *s_off, int *s_cnt, int *s_slot)
{
    const int lane = threadIdx.x & 63;
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
                const int2 vt = q.vtab[nb];
                a = vt.y;
                so = vt.x;
            }
        }
        int incl = gg_wave_incl_scan(a);
        if (nei < k3) { s_excl[nei] = M + incl - a; s_off[nei] = so; s_cnt[nei] = a; }
        M += __shfl(incl, 63, 64);
    }
    if (lane == 0) s_excl[k3] = M;
    for (int s = lane; s < P; s += 64) s_slot[s] = -1;
    __builtin_amdgcn_wave_barrier();

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
            s = gg_reservoir_pick(gg_seed(gp) + (unsigned long long)threadindex, n + 1);
        }
        if (s < P) atomicMax(&s_slot[s], id);
    }
    __builtin_amdgcn_wave_barrier();
    const int first = M > 0 ? s_slot[0] : 0;  // reference: uninitialised initID when M == 0
    for (int s = lane; s < P; s += 64) {
        row[s] = s < M ? s_slot[s] : first;
        mrow[s] = s < M ? 1.0f : 0.0f;
    }
    __builtin_amdgcn_wave_barrier();
}

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
    gg_query_up_wave(blockIdx.x, updata, up_np, Nd, gp, q, nebidx, nebmsk, s_excl, s_off, s_cnt,
                     s_slot);
}

// The shipped shapes (k = 3, P = 5: segmentation/configs/configs.yaml:99-104): ONE LANE per up point.
// The down points are sparse in the up grid (0.4-2 candidates per up point on average), so a wave
// per point leaves 60 lanes idle through three dependent lookups.  Here a lane loads its point,
// the 27 (start, population) entries of its neighbourhood (all in flight), its <= 16 candidates
// into a private LDS strip, ranks them by counting (rank = number of candidates with a smaller id)
// and resolves the reservoir in registers (P <= 8).  Points with more than 16 candidates (dense
// corners) fall to the wave-per-point routine above, one after the other, inside the same launch.
#define GG_UP_MAXC 16
template <int K>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void gg_k_query_up_lanes(const float4 *__restrict__ updata,
                                                           const int *__restrict__ up_np, int Nd,
                                                           GGGrid gp, GGQueryPtrs q, int total,
                                                           int *__restrict__ nebidx,
                                                           float *__restrict__ nebmsk)
{
    constexpr int K3 = K * K * K, HK = (K - 1) / 2;
    __shared__ int s_cand[256 * GG_UP_MAXC];
    __shared__ unsigned char s_cnei[256 * GG_UP_MAXC];
    __shared__ int s_wave[4][K3 + 1 + K3 + K3 + 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int index = blockIdx.x * 256 + tid;
    const int P = gp.P;
    bool heavy = false;
    if (index < total) {
        const int b = index / gp.O;
        const int o = index - b * gp.O;
        int *row = nebidx + (size_t)index * P;
        float *mrow = nebmsk + (size_t)index * P;
        int c3[3] = {0, 0, 0};
        int vq = -1;
        if (o < up_np[b]) {
            const float4 p = updata[index];
            vq = gg_voxel_of(p.x, p.y, p.z, gp, c3);
        }
        if (vq < 0) {  // gridify_up-inl.h:111-112 fill values
            for (int s = 0; s < P; s++) { row[s] = 0; mrow[s] = 0.0f; }
        } else {
            int2 vt[K3];
            int M = 0;
#pragma unroll
            for (int nei = 0; nei < K3; nei++) {
                const int d = nei / (K * K) - HK + c3[2];
                const int h = (nei % (K * K)) / K - HK + c3[1];
                const int w = nei % K - HK + c3[0];
                vt[nei] = make_int2(0, 0);
                if (d >= 0 && d < gp.g[2] && h >= 0 && h < gp.g[1] && w >= 0 && w < gp.g[0])
                    vt[nei] = q.vtab[(size_t)b * gp.G + (size_t)d * gp.gxy + h * gp.g[0] + w];
            }
#pragma unroll
            for (int nei = 0; nei < K3; nei++) M += vt[nei].y;
            if (M > GG_UP_MAXC) {
                heavy = true;
            } else {
                int *cand = s_cand + tid * GG_UP_MAXC;
                unsigned char *cnei = s_cnei + tid * GG_UP_MAXC;
                // positions first (LDS only), then the M <= 16 ids in rounds of eight loads: a load per (neighbour, item)
                // iteration was up to sixteen memory round trips in a row for a lane (tools/isa_chains.py: 55 loops)
                int pos = 0;
#pragma unroll
                for (int nei = 0; nei < K3; nei++)
                    for (int j = 0; j < vt[nei].y; j++) {
                        cand[pos] = vt[nei].x + j;
                        cnei[pos] = (unsigned char)nei;
                        pos++;
                    }
                // (two rounds of eight: sixteen at once cost 20 registers and a wave per SIMD; most points have M <= 8)
                const int pos0 = M > 0 ? cand[0] : 0;          // (beyond M: a valid position, its value unused)
#pragma unroll
                for (int h8 = 0; h8 < GG_UP_MAXC; h8 += 8)
                    if (M > h8) {
                        int ids[8];
#pragma unroll
                        for (int i = 0; i < 8; i++) ids[i] = q.sorted[h8 + i < M ? cand[h8 + i] : pos0];
#pragma unroll
                        for (int i = 0; i < 8; i++)
                            if (h8 + i < M) cand[h8 + i] = ids[i];
                    }
                int slot[8];
#pragma unroll
                for (int s = 0; s < 8; s++) slot[s] = -1;
                for (int i = 0; i < M; i++) {
                    const int id = cand[i];
                    int n = 0;
                    for (int j = 0; j < M; j++) n += cand[j] < id;
                    int sl = n;
                    if (n >= P) {
                        const long long threadindex = ((long long)b * Nd + id) * K3 + (K3 - 1 - cnei[i]);
                        sl = gg_reservoir_pick(gg_seed(gp) + (unsigned long long)threadindex, n + 1);
                    }
#pragma unroll
                    for (int s = 0; s < 8; s++)
                        if (s == sl) slot[s] = slot[s] > id ? slot[s] : id;   // sl < P <= 8 or no hit
                }
                const int first = M > 0 ? slot[0] : 0;  // reference: uninitialised initID when M == 0
#pragma unroll
                for (int s = 0; s < 8; s++)
                    if (s < P) {
                        row[s] = s < M ? slot[s] : first;
                        mrow[s] = s < M ? 1.0f : 0.0f;
                    }
            }
        }
    }
    unsigned long long hm = __ballot(heavy);
    int *wl = s_wave[wave];
    while (hm) {
        const int l = __builtin_ctzll(hm);
        hm &= hm - 1;
        const int idx = __shfl(index, l, 64);
        gg_query_up_wave(idx, updata, up_np, Nd, gp, q, nebidx, nebmsk, wl, wl + K3 + 1,
                         wl + 2 * K3 + 1, wl + 3 * K3 + 1);
    }
}

// ------------------------------------------------------------------------------------------
// host launchers (used by gridgcn_capi.hip)
static size_t gg_query_lds(int NC, int k3, int P)
{
    return (size_t)GG_QW * NC * (GG_QCS + 2 * k3 + 1 + 3 * P + 64) * 4;
}

int gg_launch_query_gridify(const float *data, int B, int N, const GGGrid &gp, char *wsbase,
                            const GGIndexWs &w, int *nebidx, float *nebmsk, float *cent,
                            float *centmsk, const int *centnum, hipStream_t st)
{
    GGQueryPtrs q;
    q.vtab = (const int2 *)(wsbase + w.o_vtab);
    q.sorted = (const int *)(wsbase + w.o_sorted);
    q.bkt = (const int *)(wsbase + w.o_bkt);
    q.slotfirst1 = (const int *)(wsbase + w.o_slotfirst1);
    q.centnum = centnum;
    q.exact = (const int *)(wsbase + w.o_exact);
    const long long ncent = (long long)B * gp.O;
    // centres per wave: more loads in flight per wave once there are more centres than wave slots
    int NC = 1;
    if (gp.k3 <= 64) NC = ncent > 65536 ? 4 : (ncent > 16384 ? 2 : 1);
    const int per = GG_QW * NC;
    const unsigned grid = (unsigned)B * (unsigned)((gp.O + per - 1) / per);
    const size_t lds = gg_query_lds(NC, gp.k3, gp.P);
    const float4 *d4 = (const float4 *)data;
    if (NC == 4)
        gg_k_query_gridify<4><<<grid, 64 * GG_QW, lds, st>>>(d4, N, gp, q, B, nebidx, nebmsk,
                                                             (float4 *)cent, centmsk);
    else if (NC == 2)
        gg_k_query_gridify<2><<<grid, 64 * GG_QW, lds, st>>>(d4, N, gp, q, B, nebidx, nebmsk,
                                                             (float4 *)cent, centmsk);
    else
        gg_k_query_gridify<1><<<grid, 64 * GG_QW, lds, st>>>(d4, N, gp, q, B, nebidx, nebmsk,
                                                             (float4 *)cent, centmsk);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_launch_query_up(const float *updata, const int *up_np, int B, int Nd, const GGGrid &gp,
                       char *wsbase, const GGIndexWs &w, int *nebidx, float *nebmsk,
                       hipStream_t st)
{
    GGQueryPtrs q = {};
    q.vtab = (const int2 *)(wsbase + w.o_vtab);
    q.sorted = (const int *)(wsbase + w.o_sorted);
    const long long total = (long long)B * gp.O;
    if (gp.P <= 8 && (gp.k == 3 || gp.k == 1)) {
        const unsigned grid = (unsigned)((total + 255) / 256);
        if (gp.k == 3)
            gg_k_query_up_lanes<3><<<grid, 256, 0, st>>>((const float4 *)updata, up_np, Nd, gp, q,
                                                         (int)total, nebidx, nebmsk);
        else
            gg_k_query_up_lanes<1><<<grid, 256, 0, st>>>((const float4 *)updata, up_np, Nd, gp, q,
                                                         (int)total, nebidx, nebmsk);
    } else {
        gg_k_query_up<<<B * gp.O, 64, 0, st>>>((const float4 *)updata, up_np, Nd, gp, q, nebidx,
                                               nebmsk);
    }
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
