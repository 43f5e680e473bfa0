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
// waves evaluate the batch speculatively and form its conflict matrix, then the challengers that need
// a turn (accepted, or touched by an earlier replacement of the batch) are resolved in ROUNDS -- those
// that no earlier unresolved challenger can still affect go to the 16 waves together (see "phase B"
// below).  Same verdicts as the one-by-one walk (tests/test_cas.py, bit for bit), 7x its speed.
#include "gridgcn_index.h"
#include "gridgcn_once.h"

#define GG_CAS_NT 1024

// -DGG_PROF build only (python -m grid_gcn_amd.build --prof; tools/time_cas.py --prof): where the sweep's time
// goes, accumulated by thread 0 of workgroup 0 on the 100 MHz wall clock and printed at the end of the kernel
#ifdef GG_PROF
#define GG_CAS_T(k) do { if (tid == 0 && b == 0) { const unsigned long long t_ = wall_clock64(); cas_t[k] += t_ - cas_last; cas_last = t_; } } while (0)
#define GG_CAS_N(k, v) do { if (tid == 0 && b == 0) cas_n[k] += (v); } while (0)
#else
#define GG_CAS_T(k) do {} while (0)
#define GG_CAS_N(k, v) do {} while (0)
#endif

struct GGCasArgs {
    const float4 *data;
    const int *np;
    int *slotfirst1;        // [B][O]  in/out: first point of the slot's voxel + 1
    const int *centnum;     // [B]
    const int2 *vtab;       // [B][G]  the index build's voxel table: .x = segment start in `sorted`, .y = population
    const int *sorted;      // [B*N]   point ids grouped by voxel, ascending inside a voxel
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
// the same voxel as x | y << 10 | z << 20 (what the sweep works with: no division to get back to coordinates)
__device__ __forceinline__ int gg_cas_voxel_xyz(const float4 *cloud, int i, const GGGrid &gp)
{
    const float4 p = cloud[i];
    int c[3] = {0, 0, 0};
    gg_voxel_of(p.x, p.y, p.z, gp, c);
    return c[0] | (c[1] << 10) | (c[2] << 20);
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

// CL / SL: the coverage counters / the slot arrays live in LDS (compile-time, so that their accesses are ds_*
// instructions: through a pointer that may be either LDS or global every access was a FLAT load in a branch of
// its own -- 48 of them one after the other in phase A, 4.7 us per batch of 64 challengers)
template <bool CL, bool SL>
__global__ __launch_bounds__(GG_CAS_NT) void gg_k_cas_refine(GGCasArgs a, GGGrid gp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_cas[];
    __shared__ int s_w[GG_CAS_NT / 64];
    __shared__ int s_carry;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = a.N, G = gp.G, O = gp.O, k3 = gp.k3, W = a.W;
    const int M = a.centnum[b];
    if (M < O) return;                       // fewer occupied voxels than slots: all of them are centres
    const float4 *cloud = a.data + (size_t)b * N;
    unsigned *lbm = a.bm + (size_t)b * 2 * W, *pbm = lbm + W;
    int *chal_id = a.chal + (size_t)b * 3 * N, *chal_vox = chal_id + N;
    unsigned short *cov;
    int *slotvox;
    if constexpr (CL) cov = (unsigned short *)(lds_cas + (SL ? 2 * O * 4 : 0));
    else cov = a.cov_g + (size_t)b * a.Gp;
    if constexpr (SL) slotvox = (int *)lds_cas;
    else slotvox = a.slot_g + (size_t)b * 2 * O;
    int *slotlead = slotvox + O;
    unsigned *cov32 = (unsigned *)cov;
    // counter of voxel u (u < 0: none) without a branch around the load: several of them in flight together
    auto covat = [&](int u) -> unsigned {
        const unsigned e = cov[u >= 0 ? u : 0];
        return u >= 0 ? e : 0u;
    };

#ifdef GG_PROF
    unsigned long long cas_t[6] = {0, 0, 0, 0, 0, 0}, cas_last = wall_clock64();
    int cas_n[4] = {0, 0, 0, 0};
#endif
    // ---- tables ----
    // Occupancy and the first point of every voxel come from the index build that ran just before (its voxel
    // table: start and population of a voxel's segment of the sorted ids, ascending inside a voxel -- so the first
    // entry IS the first point): no pass over the points, no atomicMin table (round 3 rebuilt both here).
    for (int w = tid; w < 2 * W; w += GG_CAS_NT) lbm[w] = 0u;
    __syncthreads();
    {
        const int2 *vt = a.vtab + (size_t)b * G;
        for (int v2 = tid; v2 < a.Gp / 2; v2 += GG_CAS_NT) {           // two voxels per thread: one 32-bit store
            unsigned w32 = 0u;
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const int v = 2 * v2 + hh;
                if (v < G) {
                    const int2 t = vt[v];
                    if (t.y > 0) {
                        w32 |= 0x8000u << (16 * hh);                   // occupied, covered by nobody yet
                        const int f = a.sorted[t.x];
                        atomicOr(&lbm[f >> 5], 1u << (f & 31));
                    }
                }
            }
            cov32[v2] = w32;
        }
    }
    for (int s = tid; s < M; s += GG_CAS_NT) {
        const int id = a.slotfirst1[(size_t)b * O + s] - 1;
        slotlead[s] = id;
        slotvox[s] = gg_cas_voxel_xyz(cloud, id, gp);           // (slot -> packed voxel coordinates)
        atomicOr(&pbm[id >> 5], 1u << (id & 31));
    }
    __syncthreads();
    for (int idx = tid; idx < M * k3; idx += GG_CAS_NT) {
        const int s = idx / k3, nei = idx - s * k3;
        const int v = slotvox[s];
        const int u = gg_cas_nb(v & 1023, (v >> 10) & 1023, v >> 20, nei, gp);
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
    for (int j = tid; j < nchal; j += GG_CAS_NT) chal_vox[j] = gg_cas_voxel_xyz(cloud, chal_id[j], gp);   // (packed)
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
    GG_CAS_T(0);
    const unsigned long long seed3 = 3ull * gg_seed(gp);
    for (int j = tid; j < nchal; j += GG_CAS_NT)
        a.chal[(size_t)b * 3 * N + 2 * N + j] =
            gg_reservoir_pick((unsigned long long)((long long)b * N + chal_id[j]) + seed3, M);
    __shared__ int sb_xyz[64], sb_sl[64], sb_id[64], sb_dec[64], sb_vixyz[64];
    __shared__ unsigned long long sb_row[64];
    __shared__ int sb_task[GG_CAS_NT / 64], sb_res[GG_CAS_NT / 64];
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
            // (no short-circuit: the nested branches of the && form were a third of phase A's instructions)
            const int d = od[i] + c2, h = oh[i] + c1, w = ow[i] + c0;
            const bool in = (lane + 64 * i < k3) & ((unsigned)d < (unsigned)gp.g[2]) &
                            ((unsigned)h < (unsigned)gp.g[1]) & ((unsigned)w < (unsigned)gp.g[0]);
            return in ? __mul24(d, gp.gxy) + __mul24(h, gp.g[0]) + w : -1;   // (full-rate multiplies: |d|, |h| < 2^11, gxy < 2^20)
        };
        // k^3 <= 32 (k = 3): BOTH windows of a challenger in one pass -- the challenger's window on lanes 0..31, the
        // incumbent's on lanes 32..63 (phase A is bound by the VALU instructions of its 16 waves: 27 useful lanes of
        // 64 and two passes per challenger were 3.3 us per batch)
        const bool k3s = k3 <= 32;
        const int neip = lane & 31, halfp = lane >> 5;
        const int odp = neip / (kk * kk) - rr, ohp = (neip % (kk * kk)) / kk - rr, owp = neip % kk - rr;
        auto nbp = [&](int pc, int pi) -> int {
            const int c = halfp ? pi : pc;
            const int d = odp + (c >> 20), h = ohp + ((c >> 10) & 1023), w = owp + (c & 1023);
            const bool in = (neip < k3) & ((unsigned)d < (unsigned)gp.g[2]) & ((unsigned)h < (unsigned)gp.g[1]) &
                            ((unsigned)w < (unsigned)gp.g[0]);
            return in ? __mul24(d, gp.gxy) + __mul24(h, gp.g[0]) + w : -1;   // (full-rate multiplies: |d|, |h| < 2^11, gxy < 2^20)
        };
        auto verdict2 = [&](unsigned e) -> bool {
            const int n0 = __popc((unsigned)__ballot(e == 0x8000u));
            const int n1 = __popc((unsigned)(__ballot(e == 0x8001u) >> 32));
            // sum of the counters of the challenger's window: a voxel lies in the windows of at most k^3 <= 32 centres
            // (the slots hold distinct voxels), so six bit planes, a ballot each -- no cross-lane adds through LDS
            const unsigned cnt = (halfp == 0 && (e & 0x8000u)) ? (e & 0x7fffu) : 0u;
            int sc = 0;
#pragma unroll
            for (int bit = 0; bit < 6; bit++) sc += __popcll(__ballot((cnt >> bit) & 1u)) << bit;
            return (float)(k3 * (n0 - n1)) > __fmul_rn(a.beta, (float)sc);
        };
        // H_add > H_rmv for challenger voxel (a0,a1,a2) against incumbent voxel (i0,i1,i2): wave-wide
        auto accept = [&](int a0, int a1, int a2, int i0, int i1, int i2) -> bool {
            if (k3s) return verdict2(covat(nbp(a0 | (a1 << 10) | (a2 << 20), i0 | (i1 << 10) | (i2 << 20))));
            int n0 = 0, n1 = 0, sc = 0;
#pragma unroll
            for (int i = 0; i < NI; i++) {
                if (i >= ni) break;
                const int uc = nbv(a0, a1, a2, i), ui = nbv(i0, i1, i2, i);
                const unsigned ec = covat(uc), ei = covat(ui);
                n0 += __popcll(__ballot(ec == 0x8000u));
                n1 += __popcll(__ballot(ei == 0x8001u));
                sc += (ec & 0x8000u) ? (int)(ec & 0x7fffu) : 0;
            }
            sc = gg_wave_sum(sc);
            return (float)(k3 * (n0 - n1)) > __fmul_rn(a.beta, (float)sc);
        };
        // A replacement writes the counters of two windows and one slot; it reads the same.  Two challengers
        // CONFLICT when a window of the one may overlap a window of the other (centres within k - 1 in every
        // dimension; clipping at the border only shrinks windows) or when they drew the same slot; everything
        // else commutes.  `row` = the conflict mask of a challenger over the batch (symmetric matrix).
        // Workgroup barrier of the sweep.  With counters and slots in LDS everything the waves share is LDS: the
        // barrier then waits for the LDS queue only and the global loads of the NEXT batch's challengers (issued
        // a batch ahead, ~1 us of latency each) stay in flight across it.
        auto sync = [&]() {
            if constexpr (CL && SL) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else __syncthreads();
        };
        auto fence = [&]() {
            if constexpr (CL && SL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else __threadfence_block();
        };
        __syncthreads();                     // (the drawn slots above are read a batch ahead, by other threads)
        int nx_vx = 0, nx_id = 0, nx_sl = 0;
        if (tid < 64 && tid < nchal) { nx_vx = chal_vox[tid]; nx_id = chal_id[tid]; nx_sl = chal_sl[tid]; }
        for (int j0 = 0; j0 < nchal; j0 += 64) {
            const int nb = nchal - j0 < 64 ? nchal - j0 : 64;
            if (tid < nb) {
                const int vx = nx_vx, sl = nx_sl;
                sb_id[tid] = nx_id;
                sb_xyz[tid] = vx;
                sb_sl[tid] = sl;
                sb_vixyz[tid] = slotvox[sl];                // the incumbent at the START of the batch
            }
            if (tid < 64 && j0 + 64 + tid < nchal) {
                nx_vx = chal_vox[j0 + 64 + tid]; nx_id = chal_id[j0 + 64 + tid]; nx_sl = chal_sl[j0 + 64 + tid];
            }
            sync();
            GG_CAS_T(1);
            GG_CAS_N(0, 1);
            // ---- phase A: speculative verdicts against the state at the start of the batch, four challengers
            // per wave with their LDS reads in flight together (one evaluation is a chain of dependent LDS
            // latencies); the same wave forms the conflict rows of its four challengers (a ballot each) ----
            {
                constexpr int U = 4;
                const int q0 = wave * U;
                if (q0 < nb) {
                    int pcs[U], pis[U];
                    bool accv[U];
#pragma unroll
                    for (int c = 0; c < U; c++) {
                        const int q = q0 + c < nb ? q0 + c : nb - 1;
                        pcs[c] = sb_xyz[q];
                        pis[c] = sb_vixyz[q];
                    }
                    if (k3s) {
                        unsigned e[U];
#pragma unroll
                        for (int c = 0; c < U; c++) e[c] = covat(nbp(pcs[c], pis[c]));
#pragma unroll
                        for (int c = 0; c < U; c++) accv[c] = verdict2(e[c]);
                    } else {
                        int n0[U], n1[U], sc[U];
#pragma unroll
                        for (int c = 0; c < U; c++) { n0[c] = 0; n1[c] = 0; sc[c] = 0; }
#pragma unroll
                        for (int i = 0; i < NI; i++) {
                            if (i >= ni) break;
                            unsigned ec[U], ei[U];
#pragma unroll
                            for (int c = 0; c < U; c++) {
                                const int uc = nbv(pcs[c] & 1023, (pcs[c] >> 10) & 1023, pcs[c] >> 20, i);
                                const int ui = nbv(pis[c] & 1023, (pis[c] >> 10) & 1023, pis[c] >> 20, i);
                                ec[c] = covat(uc);
                                ei[c] = covat(ui);
                            }
#pragma unroll
                            for (int c = 0; c < U; c++) {
                                n0[c] += __popcll(__ballot(ec[c] == 0x8000u));
                                n1[c] += __popcll(__ballot(ei[c] == 0x8001u));
                                sc[c] += (ec[c] & 0x8000u) ? (int)(ec[c] & 0x7fffu) : 0;
                            }
                        }
#pragma unroll
                        for (int c = 0; c < U; c++)
                            accv[c] = (float)(k3 * (n0[c] - n1[c])) > __fmul_rn(a.beta, (float)gg_wave_sum(sc[c]));
                    }
                    // conflict rows: the lane's two centres (biased by k - 1) against the row's two centres, which
                    // are wave uniform and go to scalar registers -- 2 VALU instructions per dimension and pair
                    const bool lv = lane < nb;
                    const int l = lv ? lane : 0;
                    const int pc_l = sb_xyz[l], pi_l = sb_vixyz[l], s_l = sb_sl[l];
                    const unsigned mm = (unsigned)(kk - 1);
                    const unsigned cx = (pc_l & 1023) + mm, cy = ((pc_l >> 10) & 1023) + mm, cz = (pc_l >> 20) + mm;
                    const unsigned ix = (pi_l & 1023) + mm, iy = ((pi_l >> 10) & 1023) + mm, iz = (pi_l >> 20) + mm;
                    auto near_s = [&](unsigned x, unsigned y, unsigned z, int sp) -> bool {
                        return ((x - (unsigned)(sp & 1023)) <= 2u * mm) & ((y - (unsigned)((sp >> 10) & 1023)) <= 2u * mm) &
                               ((z - (unsigned)(sp >> 20)) <= 2u * mm);
                    };
#pragma unroll
                    for (int c = 0; c < U; c++) {
                        const bool acc = accv[c];
                        const int q = q0 + c;
                        const int spc = __builtin_amdgcn_readfirstlane(pcs[c]), spi = __builtin_amdgcn_readfirstlane(pis[c]);
                        const int ssl = __builtin_amdgcn_readfirstlane(sb_sl[q < nb ? q : 0]);
                        const bool cf = near_s(cx, cy, cz, spc) | near_s(cx, cy, cz, spi) | near_s(ix, iy, iz, spc) |
                                        near_s(ix, iy, iz, spi) | (s_l == ssl);
                        const unsigned long long row = __ballot(lv && lane != q && q < nb && cf);
                        if (lane == 0 && q < nb) { sb_dec[q] = acc ? 1 : 0; sb_row[q] = row; }
                    }
                }
            }
            sync();
            GG_CAS_T(2);
            // ---- phase B: rounds.  Wave 0 keeps the books (lane l = challenger l): UNRESOLVED are the challengers
            // whose verdict is "accepted" but not committed yet and the DIRTY ones (an earlier replacement of the batch
            // has written what they read: to be evaluated again); a challenger is BLOCKED while an earlier
            // unresolved or blocked one conflicts with it (fixed point over the rows).  The unresolved ones that are
            // not blocked are pairwise free of conflicts and see, on everything they read, exactly the state of the
            // one-by-one walk: up to 16 of them go to the 16 waves, which evaluate (dirty) and commit together.
            // Afterwards the later challengers that conflict with a committed one become dirty.  Everyone else's
            // speculative "rejected" stands.  Same arithmetic on the same inputs: the verdicts of the walk. ----
            bool st_acc = false, st_dirty = false;
            unsigned long long row_l = 0ull;
            int bs_l = 0;
            if (wave == 0) {
                const bool valid = lane < nb;
                st_acc = valid && sb_dec[valid ? lane : 0] != 0;
                row_l = valid ? sb_row[lane] : 0ull;
                bs_l = valid ? sb_sl[lane] : -1;
            }
            while (true) {
                unsigned long long R = 0ull, UD = 0ull;
                if (wave == 0) {
                    UD = __ballot(st_dirty);
                    const unsigned long long src = UD | __ballot(st_acc);
                    const unsigned long long low = (1ull << lane) - 1ull;
                    unsigned long long bl = 0ull;
                    while (true) {
                        const unsigned long long nbl = __ballot((row_l & low & (src | bl)) != 0ull);
                        if (nbl == bl) break;
                        bl = nbl;
                    }
                    R = src & ~bl;
                    // the first 16 of R: task t -> wave t
                    unsigned long long m = R;
                    int myq = -1;
                    for (int t = 0; t < GG_CAS_NT / 64; t++) {
                        if (!m) break;
                        const int q = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        if (lane == t) myq = q;
                    }
                    R &= ~m;                                   // (beyond 16: next round)
                    if (lane < GG_CAS_NT / 64) sb_task[lane] = myq < 0 ? -1 : (myq | (((UD >> myq) & 1ull) ? 0x100 : 0));
                }
                sync();
                GG_CAS_T(3);
                if (sb_task[0] < 0) break;                     // nothing unresolved: the batch is done
                GG_CAS_N(1, 1);
                GG_CAS_N(2, __popcll(R));
                {
                    const int task = sb_task[wave];
                    if (task >= 0) {
                        const int q = task & 0xff;
                        const int pc = sb_xyz[q], s = sb_sl[q];
                        int pi = sb_vixyz[q];
                        bool acc = true;
                        if (task & 0x100) {
                            pi = slotvox[s];
                            acc = accept(pc & 1023, (pc >> 10) & 1023, pc >> 20, pi & 1023, (pi >> 10) & 1023, pi >> 20);
                        }
                        if (acc) {
                            const int a0 = pc & 1023, a1 = (pc >> 10) & 1023, a2 = pc >> 20;
                            const int i0 = pi & 1023, i1 = (pi >> 10) & 1023, i2 = pi >> 20;
                            // (the two windows of ONE replacement may overlap: the fence keeps the wave's two passes
                            //  in program order; windows of different replacements of a round are disjoint)
#pragma unroll
                            for (int i = 0; i < NI; i++) {
                                if (i >= ni) break;
                                const int ui = nbv(i0, i1, i2, i);
                                const unsigned e = covat(ui);
                                if (e & 0x8000u) cov[ui] = (unsigned short)(e - 1u);
                            }
                            fence();
#pragma unroll
                            for (int i = 0; i < NI; i++) {
                                if (i >= ni) break;
                                const int uc = nbv(a0, a1, a2, i);
                                const unsigned e = covat(uc);
                                if (e & 0x8000u) cov[uc] = (unsigned short)(e + 1u);
                            }
                            if (lane == 0) {
                                slotvox[s] = pc;
                                slotlead[s] = sb_id[q];
                            }
                        }
                        if (lane == 0) sb_res[wave] = acc ? 1 : 0;
                    }
                }
                fence();
                sync();
                GG_CAS_T(4);
                if (wave == 0) {
                    unsigned long long m = R;
                    const int res_l = sb_res[lane & (GG_CAS_NT / 64 - 1)];           // (one LDS read, not one per task)
                    for (int t = 0; m; t++) {
                        const int q = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        if (lane == q) { st_acc = false; st_dirty = false; }          // q is final
                        if (__builtin_amdgcn_readlane(res_l, t)) {
                            const bool hit = lane > q && ((row_l >> q) & 1ull);
                            st_dirty = st_dirty || hit;
                            st_acc = st_acc && !hit;
                            // later challengers on the slot that was just replaced read the NEW incumbent's window
                            // from now on: they inherit q's conflicts (a superset of their own)
                            const int s_q = __builtin_amdgcn_readlane(bs_l, q);
                            const unsigned long long ssl = __ballot(lane > q && bs_l == s_q);
                            if (ssl) {
                                const unsigned rq_lo = __builtin_amdgcn_readlane((unsigned)row_l, q);
                                const unsigned rq_hi = __builtin_amdgcn_readlane((unsigned)(row_l >> 32), q);
                                const unsigned long long row_q = ((unsigned long long)rq_hi << 32) | rq_lo;
                                if ((row_l >> q) & 1ull) row_l |= ssl;
                                if ((ssl >> lane) & 1ull) row_l |= row_q;
                                row_l &= ~(1ull << lane);
                            }
                        }
                    }
                }
            }
            sync();
        }
    }
    __syncthreads();
    for (int s = tid; s < M; s += GG_CAS_NT) a.slotfirst1[(size_t)b * O + s] = slotlead[s] + 1;
#ifdef GG_PROF
    GG_CAS_T(5);
    if (tid == 0 && b == 0)
        printf("CAS cloud 0 (us): tables %.1f | batch load %.1f | phase A %.1f | round: books %.1f, work %.1f | tail %.1f | "
               "%d challengers, %d batches, %d rounds, %d turns\n", cas_t[0] * 0.01, cas_t[1] * 0.01, cas_t[2] * 0.01,
               cas_t[3] * 0.01, cas_t[4] * 0.01, cas_t[5] * 0.01, nchal, cas_n[0], cas_n[1], cas_n[2]);
#endif
}

static size_t gg_cas_align(size_t x) { return (x + 255) & ~(size_t)255; }

size_t gg_cas_workspace_bytes(int B, int N, const GGGrid &gp)
{
    const size_t W = ((size_t)N + 31) / 32, Gp = ((size_t)gp.G + 1) & ~(size_t)1;
    return gg_cas_align((size_t)B * gp.G * 4) + gg_cas_align((size_t)B * 2 * W * 4) +
           gg_cas_align((size_t)B * 3 * N * 4) + gg_cas_align((size_t)B * Gp * 2) +
           gg_cas_align((size_t)B * 2 * gp.O * 4);
}

// vtab / sorted: what the index build of the same call left in its workspace (GGIndexWs o_vtab, o_sorted)
int gg_cas_refine(const float *data, const int *np, int B, int N, const GGGrid &gp, float beta,
                  int *slotfirst1, const int *centnum, const int2 *vtab, const int *sorted, char *ws,
                  hipStream_t st)
{
    static GGDevOnce attr_done;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)gg_k_cas_refine<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void *)gg_k_cas_refine<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess)
            return 3;
        attr_done = true;
    }
    GGCasArgs a;
    const size_t W = ((size_t)N + 31) / 32, Gp = ((size_t)gp.G + 1) & ~(size_t)1;
    a.data = (const float4 *)data; a.np = np; a.slotfirst1 = slotfirst1; a.centnum = centnum;
    char *p = ws;
    a.vtab = vtab; a.sorted = sorted;
    p += gg_cas_align((size_t)B * gp.G * 4);      // (a [B][G] table of the fast_rand query, which shares this workspace size)
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
    if (a.slot_lds >= 0 && a.cov_lds >= 0) gg_k_cas_refine<true, true><<<B, GG_CAS_NT, used, st>>>(a, gp);
    else if (a.slot_lds >= 0) gg_k_cas_refine<false, true><<<B, GG_CAS_NT, used, st>>>(a, gp);
    else if (a.cov_lds >= 0) gg_k_cas_refine<true, false><<<B, GG_CAS_NT, used, st>>>(a, gp);
    else gg_k_cas_refine<false, false><<<B, GG_CAS_NT, used, st>>>(a, gp);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
