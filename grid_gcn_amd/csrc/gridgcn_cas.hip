// gridgcn_cas.hip -- Coverage-Aware Sampling (CAS) of the centre voxels, gfx950.
//
// PARITY UNPINNED.  The reference ships CAS only inside a prebuilt binary (gridifyop/additional.so:
// ops Gridify_occaware / Gridify_occaware_s); there is no source to restate.  What is
// implemented here is the algorithm of the Grid-GCN paper (CVPR 2020, section 3.2, eq. 3-4) under a
// schedule of OUR choosing, stated in full in the tests' sequential C checker (tests/test_cas.py)
// and verified bit for bit against it -- that pins this kernel to our specification, not to the
// reference's binary.
//
//   incumbents  = the RVS sample of gridify (gridify.cu:165-189): M = min(#occupied, O) voxels
//   challengers = every occupied voxel that is not an incumbent, in order of first appearance
//   for a challenger Vc: one incumbent Vi = slot ceil(u * M) - 1, u = XORWOW(first point of Vc + 3*seed)
//       H_add = sum over occupied V in window(Vc) of  [C_V == 0] - beta * C_V / lambda
//       H_rmv = sum over occupied V in window(Vi) of  [C_V == 1]
//     C_V = number of incumbents whose k^3 window holds V, lambda = k^3; Vc replaces Vi (and the
//     counters move) when H_add > H_rmv, evaluated as  lambda*(n0 - n1) > beta * sum C_V  with
//     integer sums (one fp32 multiply, one compare: the same result on every machine).
//
// The sweep is sequential by definition (each replacement changes the counters the next challenger
// reads), so one workgroup owns a cloud: its 1024 threads build the tables in parallel -- first
// point per voxel, occupancy + coverage counters (16 bits per voxel, in LDS whenever the grid fits:
// 40^3 voxels = 128 KB of the CU's 160 KB), incumbent slots, the compacted challenger list with the
// drawn incumbent slot of every challenger -- and then walk the challengers in batches of 64: all 16
// waves evaluate the batch speculatively (a lane per window voxel: LDS latency only), wave 0 commits
// the verdicts in order and re-evaluates the few whose inputs an earlier replacement of the same
// batch has touched (see "the sweep" below): same verdicts as the one-by-one walk, ~10x its speed.
#include "gridgcn_index.h"

#define GG_CAS_NT 1024

struct GGCasArgs {
    const float4 *data;
    const int *np;
    int *slotfirst1;        // [B][O]  in/out: first point of the slot's voxel + 1
    const int *centnum;     // [B]
    int *first;             // [B][G]
    unsigned *bm;           // [B][2][W]  leader bitmap, incumbent bitmap
    int *chal;              // [B][3][N]  challenger first point, challenger voxel, drawn incumbent slot
    unsigned short *cov_g;  // [B][Gp]  (Gp = G rounded up to 2) when the counters do not fit LDS
    int *slot_g;            // [B][2][O] when the slot arrays do not fit LDS
    int N, W, Gp;
    int cov_lds, slot_lds;  // byte offsets into dynamic LDS, -1 = use the global arrays
    float beta;
};

__device__ __forceinline__ int gg_cas_voxel(const float4 *cloud, int i, const GGGrid &gp)
{
    const float4 p = cloud[i];
    return gg_voxel_of(p.x, p.y, p.z, gp, nullptr);
}

// neighbour `nei` of voxel (c0,c1,c2) in the k^3 window, -1 outside the grid (gridify.cu:240-246)
__device__ __forceinline__ int gg_cas_nb(int c0, int c1, int c2, int nei, const GGGrid &gp)
{
    const int k = gp.k, r = (k - 1) / 2;
    const int d = nei / (k * k) - r + c2;
    const int h = (nei % (k * k)) / k - r + c1;
    const int w = nei % k - r + c0;
    if (d < 0 || d >= gp.g[2] || h < 0 || h >= gp.g[1] || w < 0 || w >= gp.g[0]) return -1;
    return d * gp.gxy + h * gp.g[0] + w;
}

__global__ __launch_bounds__(GG_CAS_NT) void gg_k_cas_refine(GGCasArgs a, GGGrid gp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_cas[];
    __shared__ int s_w[GG_CAS_NT / 64];
    __shared__ int s_carry;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = a.N, G = gp.G, O = gp.O, k3 = gp.k3, W = a.W;
    const int M = a.centnum[b];
    if (M < O) return;                       // fewer occupied voxels than slots: all of them are centres
    int npts = a.np[b];
    npts = npts < 0 ? 0 : (npts > N ? N : npts);
    const float4 *cloud = a.data + (size_t)b * N;
    int *first = a.first + (size_t)b * G;
    unsigned *lbm = a.bm + (size_t)b * 2 * W, *pbm = lbm + W;
    int *chal_id = a.chal + (size_t)b * 3 * N, *chal_vox = chal_id + N;
    unsigned short *cov = a.cov_lds >= 0 ? (unsigned short *)(lds_cas + a.cov_lds)
                                         : a.cov_g + (size_t)b * a.Gp;
    int *slotvox = a.slot_lds >= 0 ? (int *)(lds_cas + a.slot_lds) : a.slot_g + (size_t)b * 2 * O;
    int *slotlead = slotvox + O;
    unsigned *cov32 = (unsigned *)cov;

    // ---- tables ----
    for (int v = tid; v < G; v += GG_CAS_NT) first[v] = 0x7fffffff;
    for (int v = tid; v < a.Gp / 2; v += GG_CAS_NT) cov32[v] = 0u;
    for (int w = tid; w < 2 * W; w += GG_CAS_NT) lbm[w] = 0u;
    __syncthreads();
    for (int i = tid; i < npts; i += GG_CAS_NT) {
        const int v = gg_cas_voxel(cloud, i, gp);
        if (v >= 0) atomicMin(&first[v], i);
    }
    __syncthreads();
    for (int v = tid; v < G; v += GG_CAS_NT) {
        const int f = first[v];
        if (f != 0x7fffffff) {
            cov[v] = 0x8000;                                   // occupied, covered by nobody yet
            atomicOr(&lbm[f >> 5], 1u << (f & 31));
        }
    }
    for (int s = tid; s < M; s += GG_CAS_NT) {
        const int id = a.slotfirst1[(size_t)b * O + s] - 1;
        slotlead[s] = id;
        slotvox[s] = gg_cas_voxel(cloud, id, gp);
        atomicOr(&pbm[id >> 5], 1u << (id & 31));
    }
    __syncthreads();
    for (int idx = tid; idx < M * k3; idx += GG_CAS_NT) {
        const int s = idx / k3, nei = idx - s * k3;
        const int v = slotvox[s];
        const int c2 = v / gp.gxy, c1 = (v - c2 * gp.gxy) / gp.g[0], c0 = v - c2 * gp.gxy - c1 * gp.g[0];
        const int u = gg_cas_nb(c0, c1, c2, nei, gp);
        if (u >= 0 && (cov[u] & 0x8000)) atomicAdd(&cov32[u >> 1], (u & 1) ? 0x10000u : 1u);
    }
    // ---- challengers: leaders that are not incumbents, ascending first point ----
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int w0 = 0; w0 < W; w0 += GG_CAS_NT) {
        const int w = w0 + tid;
        unsigned bits = w < W ? (lbm[w] & ~pbm[w]) : 0u;
        const int cnt = __popc(bits);
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        int base = s_carry;
        for (int x = 0; x < wave; x++) base += s_w[x];
        int pos = base + incl - cnt;
        while (bits) {
            const int bit = __ffs(bits) - 1;
            bits &= bits - 1;
            chal_id[pos++] = w * 32 + bit;
        }
        __syncthreads();
        if (tid == GG_CAS_NT - 1) s_carry = base + incl;
        __syncthreads();
    }
    const int nchal = s_carry;
    for (int j = tid; j < nchal; j += GG_CAS_NT) chal_vox[j] = gg_cas_voxel(cloud, chal_id[j], gp);
    __syncthreads();

    // ---- the sweep: batches of 64 challengers, evaluated speculatively by all 16 waves ----
    // The sweep is sequential by definition (an accepted challenger changes the counters the next one
    // reads), but only ~7 % are accepted and a replacement touches two k^3 windows of a 40^3 .. 64^3 grid.
    // Phase A: every wave evaluates four challengers of the batch against the state at the START of
    // the batch.  Phase B: wave 0 walks the batch in order; a challenger whose read set -- the windows
    // of its voxel and of its incumbent, its slot -- was not touched by a replacement accepted earlier
    // in the batch keeps its speculative verdict (same inputs, same arithmetic: bit-identical to the
    // sequential sweep), any other one is evaluated again on the spot.  Window overlap is tested on
    // the centres: |d|_inf <= k - 1 (clipping at the grid border only shrinks windows).
    const unsigned long long seed3 = 3ull * gg_seed(gp);
    for (int j = tid; j < nchal; j += GG_CAS_NT)
        a.chal[(size_t)b * 3 * N + 2 * N + j] =
            gg_reservoir_pick((unsigned long long)((long long)b * N + chal_id[j]) + seed3, M);
    __shared__ int sb_vx[64], sb_xyz[64], sb_sl[64], sb_id[64], sb_dec[64], sb_vixyz[64];
    const int *chal_sl = a.chal + (size_t)b * 3 * N + 2 * N;
    {
        // window offsets of this lane (nei = lane + 64*i), decoded once
        const int kk = gp.k, rr = (kk - 1) / 2;
        constexpr int NI = (GG_K3MAX + 63) / 64;
        int od[NI], oh[NI], ow[NI];
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int nei = lane + 64 * i;
            od[i] = nei / (kk * kk) - rr;
            oh[i] = (nei % (kk * kk)) / kk - rr;
            ow[i] = nei % kk - rr;
        }
        const int ni = (k3 + 63) >> 6;
        auto nbv = [&](int c0, int c1, int c2, int i) -> int {
            const int d = od[i] + c2, h = oh[i] + c1, w = ow[i] + c0;
            const bool in = lane + 64 * i < k3 && d >= 0 && d < gp.g[2] && h >= 0 && h < gp.g[1] &&
                            w >= 0 && w < gp.g[0];
            return in ? d * gp.gxy + h * gp.g[0] + w : -1;
        };
        // H_add > H_rmv for challenger voxel (a0,a1,a2) against incumbent voxel (i0,i1,i2): wave-wide
        auto accept = [&](int a0, int a1, int a2, int i0, int i1, int i2) -> bool {
            int n0 = 0, n1 = 0, sc = 0;
#pragma unroll
            for (int i = 0; i < NI; i++) {
                if (i >= ni) break;
                const int uc = nbv(a0, a1, a2, i), ui = nbv(i0, i1, i2, i);
                const unsigned ec = uc >= 0 ? cov[uc] : 0u;
                const unsigned ei = ui >= 0 ? cov[ui] : 0u;
                n0 += __popcll(__ballot(ec == 0x8000u));
                n1 += __popcll(__ballot(ei == 0x8001u));
                sc += (ec & 0x8000u) ? (int)(ec & 0x7fffu) : 0;
            }
            sc = gg_wave_sum(sc);
            return (float)(k3 * (n0 - n1)) > __fmul_rn(a.beta, (float)sc);
        };
        auto xyz_of = [&](int v) -> int {
            const int z = v / gp.gxy, y = (v - z * gp.gxy) / gp.g[0];
            return (v - z * gp.gxy - y * gp.g[0]) | (y << 10) | (z << 20);
        };
        auto near = [&](int p, int q_) -> bool {      // windows of centres p and q_ may overlap
            const int dx = (p & 1023) - (q_ & 1023), dy = ((p >> 10) & 1023) - ((q_ >> 10) & 1023),
                      dz = (p >> 20) - (q_ >> 20);
            const int m = kk - 1;
            return dx <= m && dx >= -m && dy <= m && dy >= -m && dz <= m && dz >= -m;
        };
        for (int j0 = 0; j0 < nchal; j0 += 64) {
            const int nb = nchal - j0 < 64 ? nchal - j0 : 64;
            if (tid < nb) {
                const int vx = chal_vox[j0 + tid];
                sb_id[tid] = chal_id[j0 + tid];
                sb_vx[tid] = vx;
                sb_xyz[tid] = xyz_of(vx);
                sb_sl[tid] = chal_sl[j0 + tid];
            }
            __syncthreads();
            // ---- phase A: speculative verdicts, four challengers per wave with their LDS reads in
            // flight together (one evaluation is a chain of dependent LDS latencies) ----
            {
                constexpr int U = 4;
                const int q0 = wave * U;
                if (q0 < nb) {
                    int pcs[U], pis[U], n0[U], n1[U], sc[U];
#pragma unroll
                    for (int c = 0; c < U; c++) {
                        const int q = q0 + c < nb ? q0 + c : nb - 1;
                        pcs[c] = sb_xyz[q];
                        pis[c] = xyz_of(slotvox[sb_sl[q]]);
                        n0[c] = 0; n1[c] = 0; sc[c] = 0;
                    }
#pragma unroll
                    for (int i = 0; i < NI; i++) {
                        if (i >= ni) break;
                        unsigned ec[U], ei[U];
#pragma unroll
                        for (int c = 0; c < U; c++) {
                            const int uc = nbv(pcs[c] & 1023, (pcs[c] >> 10) & 1023, pcs[c] >> 20, i);
                            const int ui = nbv(pis[c] & 1023, (pis[c] >> 10) & 1023, pis[c] >> 20, i);
                            ec[c] = uc >= 0 ? cov[uc] : 0u;
                            ei[c] = ui >= 0 ? cov[ui] : 0u;
                        }
#pragma unroll
                        for (int c = 0; c < U; c++) {
                            n0[c] += __popcll(__ballot(ec[c] == 0x8000u));
                            n1[c] += __popcll(__ballot(ei[c] == 0x8001u));
                            sc[c] += (ec[c] & 0x8000u) ? (int)(ec[c] & 0x7fffu) : 0;
                        }
                    }
#pragma unroll
                    for (int c = 0; c < U; c++) {
                        const int t = gg_wave_sum(sc[c]);
                        const bool acc = (float)(k3 * (n0[c] - n1[c])) > __fmul_rn(a.beta, (float)t);
                        if (lane == 0 && q0 + c < nb) { sb_dec[q0 + c] = acc ? 1 : 0; sb_vixyz[q0 + c] = pis[c]; }
                    }
                }
            }
            __syncthreads();
            // ---- phase B: wave 0 commits in order.  Lane l holds challenger l of the batch; only the
            // challengers that were accepted speculatively or whose inputs an accepted one has touched
            // ("dirty") need a turn of their own -- a handful per batch ----
            if (wave == 0) {
                const bool valid = lane < nb;
                const int l = valid ? lane : 0;
                const int pc_l = sb_xyz[l], s_l = sb_sl[l], pi_l = sb_vixyz[l];
                int dec_l = valid ? sb_dec[l] : 0, dirty_l = 0;
                int from = 0;
                while (true) {
                    unsigned long long m = __ballot(dec_l | dirty_l);
                    m = from < 64 ? (m >> from) << from : 0ull;
                    if (!m) break;
                    const int q = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
                    const int pc = __builtin_amdgcn_readlane(pc_l, q), s = __builtin_amdgcn_readlane(s_l, q);
                    int pi = __builtin_amdgcn_readlane(pi_l, q);
                    bool acc = __builtin_amdgcn_readlane(dec_l, q) != 0;
                    if (__builtin_amdgcn_readlane(dirty_l, q)) {
                        pi = xyz_of(slotvox[s]);
                        acc = accept(pc & 1023, (pc >> 10) & 1023, pc >> 20, pi & 1023, (pi >> 10) & 1023, pi >> 20);
                    }
                    if (acc) {
                        const int a0 = pc & 1023, a1 = (pc >> 10) & 1023, a2 = pc >> 20;
                        const int i0 = pi & 1023, i1 = (pi >> 10) & 1023, i2 = pi >> 20;
                        // (the two windows may overlap, and the next challenger reads what is written
                        // here through other lanes: fences keep the wave's accesses in program order)
#pragma unroll
                        for (int i = 0; i < NI; i++) {
                            if (i >= ni) break;
                            const int ui = nbv(i0, i1, i2, i);
                            if (ui >= 0 && (cov[ui] & 0x8000)) cov[ui] -= 1;
                        }
                        __threadfence_block();
#pragma unroll
                        for (int i = 0; i < NI; i++) {
                            if (i >= ni) break;
                            const int uc = nbv(a0, a1, a2, i);
                            if (uc >= 0 && (cov[uc] & 0x8000)) cov[uc] += 1;
                        }
                        if (lane == 0) {
                            slotvox[s] = sb_vx[q];
                            slotlead[s] = sb_id[q];
                        }
                        __threadfence_block();
                        // later challengers that read what was just written
                        if (valid && lane > q &&
                            (near(pc, pc_l) || near(pi, pc_l) || near(pc, pi_l) || near(pi, pi_l) || s_l == s))
                            dirty_l = 1;
                    }
                    if (lane == q) { dec_l = 0; dirty_l = 0; }     // q is final
                    from = q + 1;
                }
            }
            __syncthreads();
        }
    }
    __syncthreads();
    for (int s = tid; s < M; s += GG_CAS_NT) a.slotfirst1[(size_t)b * O + s] = slotlead[s] + 1;
}

static size_t gg_cas_align(size_t x) { return (x + 255) & ~(size_t)255; }

size_t gg_cas_workspace_bytes(int B, int N, const GGGrid &gp)
{
    const size_t W = ((size_t)N + 31) / 32, Gp = ((size_t)gp.G + 1) & ~(size_t)1;
    return gg_cas_align((size_t)B * gp.G * 4) + gg_cas_align((size_t)B * 2 * W * 4) +
           gg_cas_align((size_t)B * 3 * N * 4) + gg_cas_align((size_t)B * Gp * 2) +
           gg_cas_align((size_t)B * 2 * gp.O * 4);
}

int gg_cas_refine(const float *data, const int *np, int B, int N, const GGGrid &gp, float beta,
                  int *slotfirst1, const int *centnum, char *ws, hipStream_t st)
{
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)gg_k_cas_refine, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess)
            return 3;
        attr_done = true;
    }
    GGCasArgs a;
    const size_t W = ((size_t)N + 31) / 32, Gp = ((size_t)gp.G + 1) & ~(size_t)1;
    a.data = (const float4 *)data; a.np = np; a.slotfirst1 = slotfirst1; a.centnum = centnum;
    char *p = ws;
    a.first = (int *)p;            p += gg_cas_align((size_t)B * gp.G * 4);
    a.bm = (unsigned *)p;          p += gg_cas_align((size_t)B * 2 * W * 4);
    a.chal = (int *)p;             p += gg_cas_align((size_t)B * 3 * N * 4);
    a.cov_g = (unsigned short *)p; p += gg_cas_align((size_t)B * Gp * 2);
    a.slot_g = (int *)p;
    a.N = N; a.W = (int)W; a.Gp = (int)Gp; a.beta = beta;
    // LDS placement: the slot arrays first (touched by every challenger), then the counters
    const size_t budget = 153 * 1024;   // + the kernel's static LDS (batch tables: 2.6 KB) < 160 KB
    size_t used = 0;
    a.slot_lds = -1; a.cov_lds = -1;
    if ((size_t)2 * gp.O * 4 <= budget) { a.slot_lds = 0; used = (size_t)2 * gp.O * 4; }
    if (used + Gp * 2 <= budget) { a.cov_lds = (int)used; used += Gp * 2; }
    gg_k_cas_refine<<<B, GG_CAS_NT, used, st>>>(a, gp);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
