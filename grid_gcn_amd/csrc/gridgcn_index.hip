// gridgcn_index.hip -- voxel index build shared by Gridify / GridifyKNN / GridifyUp (gfx950).
//
// Replaces gridify_kernel_build_index (gridifyop/gridify.cu:102-191, gridifyknn.cu:115-204,
// gridify_up.cu:102-170).  The reference appends points to a dense [B*G, P] bucket table with
// atomics in arrival order (non-deterministic, 262-524 MB of scratch per call).  Under the
// canonical schedule S0 (threads in ascending index) a voxel's bucket is its points in ascending
// id with a "last writer wins" reservoir past P, and the centre slots are the occupied voxels in
// order of first appearance with the same reservoir past O.  So the whole build
// is a STABLE sort of the point ids by voxel, done here as a two-level split with no global
// atomics on the data path and three launches:
//
//   K1 gg_k_chunk_split  (chunk of 1024-4096 points, cloud): voxel of each point (coalesced
//      float4 stream, read once), stable split of the chunk by SLAB (a scattered sample of the
//      voxel grid, see GGSplit) in LDS; the chunk's items leave as one coalesced run per slab, 4
//      bytes per point: (voxel number inside the slab) << 12 | local point number; the
//      chunk-local exclusive slab offsets go to a small table.  Also zeroes the centre-slot array
//      and the leader bitmap, and reduces the weight statistics.
//   K2 gg_k_slab_build   (slab, cloud): gathers the slab's runs of every chunk (they are in
//      ascending point id by construction), stable split by voxel -> the sorted segment of every
//      voxel, the (start, population) table of the slab's voxels (dense, 128-byte lines, no
//      memset), the stage-1 bucket reservoir of over-full voxels (gridify.cu:145-154); the first
//      point of every occupied voxel ("leader") becomes a bit of the cloud's leader bitmap.
//   K3 gg_k_centre_slots (4096 consecutive points of a cloud, a word of the cloud's leader bitmap
//      per lane): rank of a leader among the cloud's leaders = the voxel's order of first
//      appearance = number of bitmap bits below it; RVS reservoir over the centre slots
//      (gridify.cu:165-189).  Bitmap words in, atomics out: no dependent global load.
//
// The stable split of 64 items inside a wave finds, for every lane, the set of lanes with the same
// key by one ballot per key bit; waves own contiguous ranges of the item list and keep private
// counters, so the result does not depend on any arrival order: the output is bit-identical from
// run to run and equal to schedule S0 of the reference.
#include "gridgcn_index.h"

#define GG_NT1 1024
#define GG_NW1 16
#define GG_GS 8       // lanes that copy one run together

// Slab assignment.  Voxels are taken in runs of 16 consecutive ids (128 bytes of the voxel table,
// x-neighbours mostly together); the run number u < 2^MB is scrambled by an odd multiplier (a
// bijection mod 2^MB) and the HIGH KB bits of the product pick the slab, so that every slab is a
// scattered sample of the grid and holds about the same number of points whatever the geometry
// (contiguous slabs of a plane-like cloud differ 4x in population; the tail workgroups of K2 then
// run 3x longer than the median).
struct GGSplit {
    unsigned HA, HAinv, mmask;
    int LB;       // MB - KB: bits of the run number inside a slab
    int SB;       // LB + 4: log2 of voxels per slab
    int KB;       // log2(nslab)
    int nslab, nchunk, CH;
};

__device__ __forceinline__ void gg_slab_of(int v, const GGSplit &sp, int &slab, int &vl)
{
    const unsigned h = (((unsigned)v >> GG_XRB) * sp.HA) & sp.mmask;
    slab = (int)(h >> sp.LB);
    vl = (int)(((h & ((1u << sp.LB) - 1u)) << GG_XRB) | ((unsigned)v & ((1u << GG_XRB) - 1u)));
}

// voxel id of voxel number vl of slab s (may be >= G: no such voxel)
__device__ __forceinline__ unsigned gg_voxel_of_slab(int s, int vl, const GGSplit &sp)
{
    const unsigned h = ((unsigned)s << sp.LB) | ((unsigned)vl >> GG_XRB);
    const unsigned u = (h * sp.HAinv) & sp.mmask;
    return (u << GG_XRB) | ((unsigned)vl & ((1u << GG_XRB) - 1u));
}

// exclusive prefix sum over the threads of a block of NW waves; *total = block sum.
// s_w: NW ints of LDS.  Two barriers; every thread of the block must call it.
template <int NW>
__device__ __forceinline__ int gg_block_excl_scan(int v, int *s_w, int *total)
{
    const int lane = gg_lane(), wave = (int)(threadIdx.x >> 6);
    const int incl = gg_wave_incl_scan(v);
    __syncthreads();
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const int t = s_w[w];
        if (w < wave) wbase += t;
        tot += t;
    }
    *total = tot;
    return wbase + incl - v;
}

// block sum of two ints (NW waves); s_w: 2*NW ints of LDS.  Two barriers.
template <int NW>
__device__ __forceinline__ void gg_block_sum2(int &x, int &y, int *s_w)
{
    const int lane = gg_lane(), wave = (int)(threadIdx.x >> 6);
    const int sx = gg_wave_sum(x), sy = gg_wave_sum(y);
    __syncthreads();
    if (lane == 0) { s_w[wave] = sx; s_w[NW + wave] = sy; }
    __syncthreads();
    int tx = 0, ty = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) { tx += s_w[w]; ty += s_w[NW + w]; }
    x = tx;
    y = ty;
}

// lanes of the wave whose key equals mine (valid lanes only); one ballot per key bit.
__device__ __forceinline__ unsigned long long gg_wave_peers(bool valid, unsigned key, int nbits)
{
    unsigned long long peers = __ballot(valid);
    for (int k = 0; k < nbits; k++) {
        const bool bit = (key >> k) & 1u;
        const unsigned long long bal = __ballot(valid && bit);
        peers &= bit ? bal : ~bal;
    }
    return peers;
}

// ------------------------------------------------------------------------------------------
// K1.  grid (nchunk, B), block 1024, chunk = 1024*IPT points, dynamic LDS = (16*nslab + chunk)
// ints.  Wave w owns the points [w*64*IPT, (w+1)*64*IPT) of the chunk, 64 at a time: order of
// (wave, iteration, lane) = ascending point id, which the split preserves inside every slab.
template <int IPT>
__global__ __launch_bounds__(GG_NT1) void gg_k_chunk_split(
    const float4 *__restrict__ data, const int *__restrict__ np, int N, int B, GGGrid gp, GGSplit sp,
    unsigned *__restrict__ part, int *__restrict__ ctab, unsigned long long *__restrict__ wsum_blk,
    int *__restrict__ zero_base, int zero_words)
{
    constexpr int CH = GG_NT1 * IPT;
    extern __shared__ __attribute__((aligned(16))) int lds1[];
    const int nslab = sp.nslab;
    int *wc = lds1;                                         // [16][nslab]
    unsigned *stage = (unsigned *)(lds1 + GG_NW1 * nslab);  // [CH]
    __shared__ int s_w[GG_NW1];
    __shared__ long long s_sum[GG_NW1];
    __shared__ int s_flag[GG_NW1];
    const int nchunk = sp.nchunk;
    int b, chunk;
    gg_cloud_item(blockIdx.x, nchunk, B, b, chunk);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wgid = b * nchunk + chunk;
    GG_STAMP(0, wgid, 0);

    // the point loads do not wait for anything (the valid-count mask is applied afterwards)
    float4 p[IPT];
#pragma unroll
    for (int j = 0; j < IPT; j++) {
        const int ip = chunk * CH + wave * (64 * IPT) + j * 64 + lane;
        p[j] = ip < N ? data[(size_t)b * N + ip] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    int nvalid = np[b];
    nvalid = nvalid < N ? nvalid : N;
    for (int j = tid; j < GG_NW1 * nslab; j += GG_NT1) wc[j] = 0;
    if (zero_words > 0) {  // centre slots, zeroed for K3 (stream order)
        const int nwg = nchunk * B;
        const int per = (zero_words + nwg - 1) / nwg;
        const int z0 = wgid * per, z1 = (z0 + per < zero_words) ? z0 + per : zero_words;
        for (int j = z0 + tid; j < z1; j += GG_NT1) zero_base[j] = 0;
    }
    __syncthreads();
    GG_STAMP(0, wgid, 1);

    int slab[IPT], vl[IPT];
    long long aw = 0;
    int flags = 0;  // bit 0: non-integer / huge weight, bit 1: weight != 1
#pragma unroll
    for (int j = 0; j < IPT; j++) {
        const int ip = chunk * CH + wave * (64 * IPT) + j * 64 + lane;
        slab[j] = -1;
        vl[j] = 0;
        if (ip < nvalid) {
            const int v = gg_voxel_of(p[j].x, p[j].y, p[j].z, gp, nullptr);
            if (v >= 0) {
                const float w = p[j].w;
                const bool bad = !(truncf(w) == w) || !(fabsf(w) < 8388608.0f);
                flags |= (bad ? 1 : 0) | ((w == 1.0f) ? 0 : 2);
                aw += bad ? 0 : (long long)fabsf(w);
                gg_slab_of(v, sp, slab[j], vl[j]);
                atomicAdd(&wc[wave * nslab + slab[j]], 1);
            }
        }
    }
    {   // weight statistics of the chunk (finished by thread 1023 behind the next barrier)
        const long long ws = gg_wave_sum_ll(aw);
        int f = flags;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) f |= __shfl_xor(f, d, 64);
        if (lane == 0) { s_sum[wave] = ws; s_flag[wave] = f; }
    }
    GG_STAMP(0, wgid, 2);
    __syncthreads();
    if (tid == GG_NT1 - 1) {
        long long t = 0;
        int ff = 0;
#pragma unroll
        for (int w = 0; w < GG_NW1; w++) { t += s_sum[w]; ff |= s_flag[w]; }
        wsum_blk[(size_t)b * nchunk + chunk] = (unsigned long long)t |
                                              ((ff & 1) ? (1ull << 63) : 0ull) |
                                              ((ff & 2) ? (1ull << 62) : 0ull);
    }
    // per slab: exclusive prefix over the waves (in registers), then over the slabs
    int c[GG_NW1];
    int tot = 0;
    if (tid < nslab) {
#pragma unroll
        for (int w = 0; w < GG_NW1; w++) c[w] = wc[w * nslab + tid];
#pragma unroll
        for (int w = 0; w < GG_NW1; w++) { const int t = c[w]; c[w] = tot; tot += t; }
    }
    int total;
    const int excl = gg_block_excl_scan<GG_NW1>(tot, s_w, &total);
    int *row = ctab + ((size_t)b * nchunk + chunk) * (nslab + 1);
    if (tid < nslab) {
        row[tid] = excl;
#pragma unroll
        for (int w = 0; w < GG_NW1; w++) wc[w * nslab + tid] = c[w] + excl;
    }
    if (tid == 0) row[nslab] = total;
    __syncthreads();
    GG_STAMP(0, wgid, 3);
    // stable placement
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < IPT; j++) {
        const bool valid = slab[j] >= 0;
        const unsigned key = valid ? (unsigned)slab[j] : 0u;
        const unsigned long long peers = gg_wave_peers(valid, key, sp.KB);
        if (valid) {
            const int rank = __popcll(peers & lt), npeer = __popcll(peers);
            const int base = wc[wave * nslab + key];
            const unsigned q = (unsigned)(wave * (64 * IPT) + j * 64 + lane);
            stage[base + rank] = ((unsigned)vl[j] << 12) | q;
            if (IPT > 1 && rank == npeer - 1) wc[wave * nslab + key] = base + npeer;
        }
    }
    GG_STAMP(0, wgid, 4);
    __syncthreads();
    unsigned *dst = part + (size_t)b * N + (size_t)chunk * CH;
    for (int t = tid; t < total; t += GG_NT1) dst[t] = stage[t];
    GG_STAMP(0, wgid, 5);
}

// ------------------------------------------------------------------------------------------
// K2.  grid (nslab, B), block 64*NW, dynamic LDS:
//   wc[NW][S] | voff[S+1] | ltmp[S] | roff[nchunk+1] | rsrc[nchunk] | lid[NW][CAPW] |
//   lvl[NW][CAPW] (u16)
// The slab's item list = its runs in ascending chunk order (ascending point id).  Wave w owns the
// runs of the chunks [nchunk*w/NW, nchunk*(w+1)/NW) and streams that part of the list through its
// private LDS tile (CAPW items at a time; one tile in the common case, then the second pass
// reuses it).  Runs are copied by groups of 8 lanes, two runs in flight per group.
struct GGSlabArgs {
    const unsigned *part;
    const int *ctab;
    int2 *vtab;
    int *sorted, *bkt;
    unsigned *lbm;   // leader bitmap [B][ceil(N/32)]: bit i = point i is the first point of its voxel
    int N;
};

template <int NW> struct GGCapW { static constexpr int value = NW >= 8 ? 256 : 512; };

template <bool WITH_CENTRES, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_num_sgpr(80))) void gg_k_slab_build(GGSlabArgs a, int B, GGGrid gp, GGSplit sp)
{
    constexpr int NT = 64 * NW;
    constexpr int CAPW = GGCapW<NW>::value;
    extern __shared__ __attribute__((aligned(16))) int lds2[];
    const int S = 1 << sp.SB, nchunk = sp.nchunk, N = a.N, CH = sp.CH;
    int *wc = lds2;                      // [NW][S]
    int *voff = wc + NW * S;             // [S+1]
    int *ltmp = voff + S + 1;            // [S]
    int *roff = ltmp + S;                // [nchunk+1]
    int *rsrc = roff + nchunk + 1;       // [nchunk]
    int *lid = rsrc + nchunk;            // [NW][CAPW]
    unsigned short *lvl = (unsigned short *)(lid + NW * CAPW);
    __shared__ int s_w[2 * NW];
    __shared__ int s_dense;
    int b, s;
    gg_cloud_item(blockIdx.x, sp.nslab, B, b, s);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wgid = b * sp.nslab + s;
    GG_STAMP(1, wgid, 0);

    // ---- runs of this slab: list offsets, and the slab's base = points in lower slabs ----
    int n_s = 0, base_s = 0;
    for (int c0 = 0; c0 < nchunk; c0 += NT) {
        const int c = c0 + tid;
        int st = 0, len = 0;
        if (c < nchunk) {
            const int *row = a.ctab + ((size_t)b * nchunk + c) * (sp.nslab + 1);
            st = row[s];
            len = row[s + 1] - st;
        }
        if (c0 == 0) {  // overlapped with the table loads
            for (int j = tid; j < NW * S; j += NT) wc[j] = 0;
            if (tid == 0) s_dense = 0;
        }
        int tl;
        const int ex = gg_block_excl_scan<NW>(len, s_w, &tl);
        if (c < nchunk) {
            roff[c] = n_s + ex;
            rsrc[c] = c * CH + st - (n_s + ex);
        }
        n_s += tl;
        base_s += st;  // per thread; summed below
    }
    {
        int dummy = 0;
        gg_block_sum2<NW>(base_s, dummy, s_w);
    }
    if (tid == 0) roff[nchunk] = n_s;
    __syncthreads();
    GG_STAMP(1, wgid, 1);

    const int cw0 = (nchunk * wave) / NW, cw1 = (nchunk * (wave + 1)) / NW;
    const int L0 = roff[cw0], L1 = roff[cw1];
    const int ntile = (L1 - L0 + CAPW - 1) / CAPW;  // per wave
    int *mylid = lid + wave * CAPW;
    unsigned short *mylvl = lvl + wave * CAPW;
    const unsigned *cloudpart = a.part + (size_t)b * N;
    const int grp = lane / GG_GS, gl = lane % GG_GS;

    // stage the list positions [T0, T1) of this wave: group g copies the runs cw0+g, cw0+g+8, ...
    auto gather_tile = [&](int T0, int T1) {
        for (int c = cw0 + grp; c < cw1; c += 2 * (64 / GG_GS)) {
            const int c2 = c + 64 / GG_GS;
            const int r0 = roff[c], r1 = roff[c + 1];
            int q0 = 0, q1 = 0;
            if (c2 < cw1) { q0 = roff[c2]; q1 = roff[c2 + 1]; }
            const int lo0 = r0 > T0 ? r0 : T0, hi0 = r1 < T1 ? r1 : T1;
            const int lo1 = q0 > T0 ? q0 : T0, hi1 = q1 < T1 ? q1 : T1;
            int pa = lo0 + gl, pb = lo1 + gl;
            while (pa < hi0 || pb < hi1) {  // two runs in flight per group
                unsigned ia = 0u, ib = 0u;
                if (pa < hi0) ia = cloudpart[rsrc[c] + pa];
                if (pb < hi1) ib = cloudpart[rsrc[c2] + pb];
                if (pa < hi0) {
                    mylid[pa - T0] = c * CH + (int)(ia & 4095u);
                    mylvl[pa - T0] = (unsigned short)(ia >> 12);
                }
                if (pb < hi1) {
                    mylid[pb - T0] = c2 * CH + (int)(ib & 4095u);
                    mylvl[pb - T0] = (unsigned short)(ib >> 12);
                }
                pa += GG_GS;
                pb += GG_GS;
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    // ---- pass 1: per-wave voxel populations ----
    for (int t = 0; t < ntile; t++) {
        const int T0 = L0 + t * CAPW, T1 = (T0 + CAPW < L1) ? T0 + CAPW : L1;
        gather_tile(T0, T1);
        for (int i = lane; i < T1 - T0; i += 64) atomicAdd(&wc[wave * S + mylvl[i]], 1);
        __builtin_amdgcn_wave_barrier();
    }
    GG_STAMP(1, wgid, 2);
    __syncthreads();
    GG_STAMP(1, wgid, 3);

    // ---- voxel offsets: thread t owns the VPT consecutive voxels [t*VPT, (t+1)*VPT) ----
    const int VPT = (S + NT - 1) / NT;
    const int j0 = tid * VPT;
    int mine = 0;
    for (int j = j0; j < j0 + VPT && j < S; j++) {
        int run = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const int cw = wc[w * S + j];
            wc[w * S + j] = run;
            run += cw;
        }
        ltmp[j] = run;  // population
        mine += run;
    }
    int tot_chk;
    int run0 = gg_block_excl_scan<NW>(mine, s_w, &tot_chk);
    for (int j = j0; j < j0 + VPT && j < S; j++) {
        const int cj = ltmp[j];
        voff[j] = run0;
#pragma unroll
        for (int w = 0; w < NW; w++) wc[w * S + j] += run0;
        run0 += cj;
    }
    if (tid == NT - 1) voff[S] = run0;
    __syncthreads();
    // voxel table of the slab's voxels (coalesced: 16 consecutive voxel numbers = 16 consecutive
    // voxel ids = one 128-byte line), -1 fill of the bucket of over-full voxels
    const size_t gbase = (size_t)b * N + base_s;  // absolute start of the slab in sorted/bkt/lead
    bool dense = false;
    for (int j = tid; j < S; j += NT) {
        const unsigned v = gg_voxel_of_slab(s, j, sp);
        if (v < (unsigned)gp.G) {
            const int vo = voff[j], cj = voff[j + 1] - vo;
            a.vtab[(size_t)b * gp.G + v] = make_int2((int)(gbase + vo), cj);
            if (WITH_CENTRES && cj > gp.P) {
                dense = true;
                for (int q = 0; q < gp.P; q++) a.bkt[gbase + vo + q] = -1;
            }
        }
    }
    if (WITH_CENTRES) {
        if (dense) s_dense = 1;
        __syncthreads();
        if (s_dense) {
            // the -1 fills above must be in L2 before other waves' atomicMax on the same words
            __threadfence();
            __syncthreads();
        }
    }
    GG_STAMP(1, wgid, 4);

    // ---- pass 2: stable placement ----
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int t = 0; t < ntile; t++) {
        const int T0 = L0 + t * CAPW, T1 = (T0 + CAPW < L1) ? T0 + CAPW : L1;
        if (ntile > 1) gather_tile(T0, T1);
        const int len = T1 - T0;
        for (int i0 = 0; i0 < len; i0 += 64) {
            const int i = i0 + lane;
            const bool valid = i < len;
            const unsigned vl = valid ? (unsigned)mylvl[i] : 0u;
            const int id = valid ? mylid[i] : 0;
            const unsigned long long peers = gg_wave_peers(valid, vl, sp.SB);
            const int rank = __popcll(peers & lt), npeer = __popcll(peers);
            // (every peer reads the voxel's counter, THEN the last peer advances it: one load instruction, one later
            //  store of the wave -- lockstep; the store depends on the load's value)
            const int base = valid ? wc[wave * S + vl] : 0;
            GG_LOCKSTEP();
            if (valid) {
                const int pos = base + rank;
                if (rank == npeer - 1) wc[wave * S + vl] = base + npeer;
                a.sorted[gbase + pos] = id;
                if (WITH_CENTRES) {
                    const int vo = voff[vl];
                    const int n = pos - vo;            // rank of the point inside its voxel
                    const int cv = voff[vl + 1] - vo;  // population of the voxel
                    if (n == 0)   // first point of its voxel: a leader of the cloud
                        atomicOr(&a.lbm[(size_t)b * ((N + 31) >> 5) + (id >> 5)], 1u << (id & 31));
                    if (cv > gp.P) {
                        // S0: item n < P sits in slot n; item n >= P overwrites slot r(n) if
                        // r(n) < P (gridify.cu:146-153).  Last writer = largest n = largest id.
                        int sl = n;
                        if (n >= gp.P) {
                            const int gi = (int)((long long)b * N + id);
                            sl = gg_reservoir_pick((unsigned long long)(long long)gi + gg_seed(gp),
                                                   n + 1);
                        }
                        if (sl < gp.P) atomicMax(&a.bkt[gbase + vo + sl], id);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    GG_STAMP(1, wgid, 5);
}

// ------------------------------------------------------------------------------------------
// K3.  grid = B * ceil(nw32 / 128) workgroups of 128 threads; a thread owns ONE word of the cloud's
// leader bitmap (32 points).  Centre slots = RVS reservoir over the occupied voxels in order of
// first appearance (gridify.cu:165-189): t = rank of the voxel's first point among all first
// points = number of bitmap bits below it.  Voxel t < O sits in slot t; voxel t >= O overwrites
// slot r(t) if r(t) < O; "last writer wins" of S0 = largest t = largest first point -> atomicMax on
// (first point id + 1), 0 = empty.
// Per wave: the set bits are compacted into an LDS list (position = rank), then the list is walked
// 64 leaders at a time with every lane busy.  No dependent global load anywhere: bitmap words in,
// atomics out.  (A form that also fetched the landing leaders' points to record the voxel id in the
// slot -- one dependent load less in the query -- cost the first workgroup of every cloud, where all
// ~950 leaders land, 8 us for its scattered loads: dropped.)
#define GG_NT3 128
__global__ __launch_bounds__(GG_NT3) void gg_k_centre_slots(
    int N, int B, GGGrid gp, int nchunk, const unsigned *__restrict__ lbm,
    const unsigned long long *__restrict__ wsum_blk, int *__restrict__ slotfirst1,
    int *__restrict__ centnum, int *__restrict__ exact)
{
    constexpr int NW = GG_NT3 / 64;
    __shared__ int s_before[NW], s_wtot[NW];
    __shared__ int s_list[NW][2048];       // leader ids of a wave's 64 words
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nw32 = (N + 31) >> 5;
    const int nwg = (nw32 + GG_NT3 - 1) / GG_NT3;
    int b, r;
    gg_cloud_item(blockIdx.x, nwg, B, b, r);
    const unsigned *bm = lbm + (size_t)b * nw32;
    const int w_first = r * GG_NT3;
    GG_STAMP(2, blockIdx.x, 0);
    // ---- leaders below this workgroup's words (unconditional loads from clamped addresses, the
    //      select applied at use: eight in flight per thread) ----
    const int wi = w_first + tid;
    const unsigned word = bm[wi < nw32 ? wi : nw32 - 1] & (wi < nw32 ? ~0u : 0u);
    int before = 0;
    for (int j0 = tid; j0 < w_first; j0 += 8 * GG_NT3) {
        unsigned x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int j = j0 + u * GG_NT3;
            x[u] = bm[j < w_first ? j : 0];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) before += j0 + u * GG_NT3 < w_first ? __popc(x[u]) : 0;
    }
    before = gg_wave_sum(before);
    const int pc = __popc(word);
    const int incl = gg_wave_incl_scan(pc);
    const int wtot = __shfl(incl, 63, 64);
    if (lane == 0) { s_before[wave] = before; s_wtot[wave] = wtot; }
    __syncthreads();
    int nbefore = 0, below_waves = 0, wg_total = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        nbefore += s_before[w];
        if (w < wave) below_waves += s_wtot[w];
        wg_total += s_wtot[w];
    }
    const int t_wave = nbefore + below_waves;   // rank of this wave's first leader
    GG_STAMP(2, blockIdx.x, 1);
    // ---- compact this wave's leaders: list position = rank - t_wave ----
    {
        int q = incl - pc;
        unsigned m = word;
        const int id0 = wi << 5;
        while (m) {
            s_list[wave][q++] = id0 + __builtin_ctz(m);
            m &= m - 1u;
        }
    }
    __builtin_amdgcn_wave_barrier();
    GG_STAMP(2, blockIdx.x, 2);
    const int O = gp.O;
    const unsigned long long seed2 = 2ull * gg_seed(gp);
    int *slots = slotfirst1 + (size_t)b * O;
    for (int i = lane; i < wtot; i += 64) {
        const int id = s_list[wave][i];
        const int t = t_wave + i;
        int sl = t;
        if (t >= O) {
            const int gi = (int)((long long)b * N + id);
            sl = gg_reservoir_pick((unsigned long long)(long long)gi + seed2, t + 1);
        }
        if (sl < O) atomicMax(&slots[sl], id + 1);
    }
    GG_STAMP(2, blockIdx.x, 3);
    if (r == nwg - 1 && wave == 0) {
        // the cloud's last words: all leaders counted.  Weights of the cloud are integers and
        // sum(|w|) < 2^23: every partial sum of S0's total_weight accumulation is exact, so it may
        // be evaluated in any order
        const int nlead = nbefore + wg_total;
        long long ws = 0;
        int fl = 0;
        for (int j = lane; j < nchunk; j += 64) {
            const unsigned long long x = wsum_blk[(size_t)b * nchunk + j];
            fl |= (int)(x >> 62);
            ws += (long long)(x & ~(3ull << 62));
        }
        ws = gg_wave_sum_ll(ws);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) fl |= __shfl_xor(fl, d, 64);
        if (lane == 0) {
            centnum[b] = nlead < O ? nlead : O;
            const int ex = (!(fl & 2) && ws < (1ll << 23)) ? 1 : 0;
            exact[b] = ex | ((ex && !(fl & 1)) ? 2 : 0);
        }
    }
}
GG_PROF_SETTER(gridgcn_prof_set_index)

// ------------------------------------------------------------------------------------------
// Small clouds (N <= 4096 points: layers 1.. of every model, the 1024-point classification input).
// The three launches above pay three launch ramps and two kernel boundaries for a few hundred KB; here
// the whole build is ONE launch of one 1024-thread workgroup per cloud, everything in LDS:
//   1. voxel of every point (list position = point id);
//   2. stable LSD radix sort of the list by voxel id, <= 8 bits per pass (1-3 passes for G < 2^24),
//      each pass the same per-wave-counter + ballot split as K1 -- waves own contiguous ranges of the
//      list, so the order inside a digit never depends on arrival; invalid points drop out in pass 0;
//   3. segment heads by comparing neighbours of the sorted list: start, population and every item's
//      rank inside its voxel from ONE block scan of the head flags; the head of a segment is the
//      voxel's first point (leader);
//   4. voxel table (zero fill issued at kernel start, heads overwrite behind a drained barrier),
//      sorted ids, leader bitmap -> rank of first appearance (popcount prefix) -> RVS centre slots,
//      stage-1 bucket reservoir of over-full voxels -- LDS atomicMax, written out once.
// Produces exactly what K1-K3 leave in the workspace (vtab, sorted, bkt, slotfirst1, centnum, exact):
// the query kernels do not know which build ran.  Same arithmetic and the same "largest id wins"
// rule as above, so the result is the S0 result bit for bit (tests: every golden / fuzz case with
// N <= 4096 runs through this kernel, and test_small_build_equals_split_build compares the two
// builds' outputs on the same inputs).
#define GG_SM_MAXN 4096
#define GG_SM_MAXO 4096
#define GG_SM_NW 16

struct GGSmallArgs {
    const float4 *data;
    const int *np;
    int2 *vtab;
    int *sorted, *bkt, *slotfirst1, *centnum, *exact;
    int N, B, nbits;
};

static size_t gg_small_lds(bool with_centres)
{
    // vox[2][4096] | id[2][4096] u16 | wc[16][256] | H[4097] | (bkt[4096] | slot[4096] | bm[128] | pre[128])
    size_t n = 2 * GG_SM_MAXN * 4 + 2 * GG_SM_MAXN * 2 + GG_SM_NW * 256 * 4 + (GG_SM_MAXN + 8) * 4;
    if (with_centres) n += GG_SM_MAXN * 4 + GG_SM_MAXO * 4 + 256 * 4;
    return n;
}

template <int IPT, bool WITH_CENTRES>
__global__ __launch_bounds__(1024) void gg_k_small_build(GGSmallArgs a, GGGrid gp)
{
    constexpr int CHN = 1024 * IPT;
    extern __shared__ __attribute__((aligned(16))) int ldss[];
    int *vox0 = ldss, *vox1 = vox0 + GG_SM_MAXN;
    unsigned short *id0 = (unsigned short *)(vox1 + GG_SM_MAXN), *id1 = id0 + GG_SM_MAXN;
    int *wc = (int *)(id1 + GG_SM_MAXN);            // [16][256]
    int *H = wc + GG_SM_NW * 256;                   // [nheads + 1] list positions of the segment heads
    int *s_bkt = H + GG_SM_MAXN + 8;                // [N]   (centres)
    int *s_slot = s_bkt + GG_SM_MAXN;               // [O]
    unsigned *s_bm = (unsigned *)(s_slot + GG_SM_MAXO);   // [128] leader bitmap
    int *s_pre = (int *)(s_bm + 128);               // [128] leaders below a word
    __shared__ int s_w[GG_SM_NW];
    __shared__ long long s_sum[GG_SM_NW];
    __shared__ int s_flag[GG_SM_NW];
    const int b = blockIdx.x, N = a.N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long seed = gg_seed(gp);

    float4 p[IPT];
#pragma unroll
    for (int j = 0; j < IPT; j++) {
        const int ip = wave * (64 * IPT) + j * 64 + lane;
        p[j] = ip < N ? a.data[(size_t)b * N + ip] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    int nvalid = a.np[b];
    nvalid = nvalid < N ? nvalid : N;
    // voxel table: empty everywhere first (the heads overwrite their entries in phase 4)
    int2 *vt = a.vtab + (size_t)b * gp.G;
    for (int v = tid; v < gp.G; v += 1024) vt[v] = make_int2(0, 0);

    // ---- 1. voxels; weight statistics of the cloud (as K1 / K3) ----
    {
        long long aw = 0;
        int flags = 0;
#pragma unroll
        for (int j = 0; j < IPT; j++) {
            const int ip = wave * (64 * IPT) + j * 64 + lane;
            int v = -1;
            if (ip < nvalid) {
                v = gg_voxel_of(p[j].x, p[j].y, p[j].z, gp, nullptr);
                if (v >= 0) {
                    const float w = p[j].w;
                    const bool bad = !(truncf(w) == w) || !(fabsf(w) < 8388608.0f);
                    flags |= (bad ? 1 : 0) | ((w == 1.0f) ? 0 : 2);
                    aw += bad ? 0 : (long long)fabsf(w);
                }
            }
            vox0[ip] = v;
            id0[ip] = (unsigned short)ip;
        }
        if (WITH_CENTRES) {
            const long long ws = gg_wave_sum_ll(aw);
            int f = flags;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) f |= __shfl_xor(f, d, 64);
            if (lane == 0) { s_sum[wave] = ws; s_flag[wave] = f; }
        }
    }

    // ---- 2. stable LSD radix sort by voxel id ----
    const int npass = a.nbits <= 8 ? 1 : (a.nbits <= 16 ? 2 : 3);
    int rb = (a.nbits + npass - 1) / npass;
    rb = rb < 1 ? 1 : rb;
    const int nb = 1 << rb;
    int *cv = vox0, *nv = vox1;
    unsigned short *ci = id0, *ni = id1;
    int nlist = CHN;                     // pass 0: validity is v >= 0; later: position < nlist
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int pass = 0; pass < npass; pass++) {
        const int shift = pass * rb;
        for (int j = tid; j < GG_SM_NW * nb; j += 1024) wc[j] = 0;
        __syncthreads();
        int v[IPT], dig[IPT];
        unsigned short idv[IPT];
        bool val[IPT];
#pragma unroll
        for (int j = 0; j < IPT; j++) {
            const int pos = wave * (64 * IPT) + j * 64 + lane;
            v[j] = cv[pos];
            idv[j] = ci[pos];
            val[j] = pass == 0 ? v[j] >= 0 : pos < nlist;
            dig[j] = (v[j] >> shift) & (nb - 1);
            if (val[j]) atomicAdd(&wc[wave * nb + dig[j]], 1);
        }
        __syncthreads();
        int c[GG_SM_NW];
        int tot = 0;
        if (tid < nb) {
#pragma unroll
            for (int w = 0; w < GG_SM_NW; w++) c[w] = wc[w * nb + tid];
#pragma unroll
            for (int w = 0; w < GG_SM_NW; w++) { const int t = c[w]; c[w] = tot; tot += t; }
        }
        int total;
        const int excl = gg_block_excl_scan<GG_SM_NW>(tot, s_w, &total);
        if (tid < nb) {
#pragma unroll
            for (int w = 0; w < GG_SM_NW; w++) wc[w * nb + tid] = c[w] + excl;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < IPT; j++) {
            const unsigned long long peers = gg_wave_peers(val[j], (unsigned)dig[j], rb);
            const int rank = __popcll(peers & lt), npeer = __popcll(peers);
            const int base = val[j] ? wc[wave * nb + dig[j]] : 0;     // (all peers read, then the last one advances)
            GG_LOCKSTEP();
            if (val[j]) {
                nv[base + rank] = v[j];
                ni[base + rank] = idv[j];
                if (IPT > 1 && rank == npeer - 1) wc[wave * nb + dig[j]] = base + npeer;
            }
        }
        nlist = total;
        __syncthreads();
        { int *t = cv; cv = nv; nv = t; }
        { unsigned short *t = ci; ci = ni; ni = t; }
    }
    const int n = nlist;                 // in-grid points of the cloud, cv / ci sorted by (voxel, id)

    // ---- 3. segments: thread t owns the list positions [t*IPT, (t+1)*IPT) ----
    int *hr = nv;                        // head rank (= segment number) of every list position
    bool head[IPT];
    int nh = 0;
#pragma unroll
    for (int j = 0; j < IPT; j++) {
        const int pos = tid * IPT + j;
        head[j] = pos < n && (pos == 0 || cv[pos] != cv[pos - 1]);
        nh += head[j] ? 1 : 0;
    }
    int nheads;
    int run = gg_block_excl_scan<GG_SM_NW>(nh, s_w, &nheads);
#pragma unroll
    for (int j = 0; j < IPT; j++) {
        const int pos = tid * IPT + j;
        if (head[j]) H[run++] = pos;
        if (pos < n) hr[pos] = run - 1;
    }
    if (tid == 0) H[nheads] = n;
    if (WITH_CENTRES) {
        if (tid < 128) s_bm[tid] = 0u;
        for (int o = tid; o < gp.O; o += 1024) s_slot[o] = 0;
    }
    // (the zero fill of the voxel table must have reached L2 before another wave's head entry)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- 4. outputs ----
    const size_t gbase = (size_t)b * N;
    int st_[IPT], pop_[IPT], id_[IPT];
#pragma unroll
    for (int j = 0; j < IPT; j++) {
        const int pos = tid * IPT + j;
        st_[j] = 0; pop_[j] = 0; id_[j] = 0;
        if (pos < n) {
            const int r = hr[pos];
            st_[j] = H[r];
            pop_[j] = H[r + 1] - st_[j];
            id_[j] = (int)ci[pos];
            a.sorted[gbase + pos] = id_[j];
            if (head[j]) {
                vt[cv[pos]] = make_int2((int)(gbase + pos), pop_[j]);
                if (WITH_CENTRES) atomicOr(&s_bm[id_[j] >> 5], 1u << (id_[j] & 31));
            }
            if (WITH_CENTRES && pop_[j] > gp.P && pos - st_[j] < gp.P) s_bkt[pos] = -1;
        }
    }
    if (!WITH_CENTRES) return;
    __syncthreads();
    {
        const int pc = tid < 128 ? __popc(s_bm[tid]) : 0;
        int nlead;
        const int ex = gg_block_excl_scan<GG_SM_NW>(pc, s_w, &nlead);
        if (tid < 128) s_pre[tid] = ex;
    }
    __syncthreads();
    const int O = gp.O, P = gp.P;
#pragma unroll
    for (int j = 0; j < IPT; j++) {
        const int pos = tid * IPT + j;
        if (pos < n) {
            const int id = id_[j];
            const int gi = (int)((long long)b * N + id);
            if (head[j]) {
                // rank of first appearance of the voxel = leaders with a smaller id (gridify.cu:165-189)
                const int t = s_pre[id >> 5] + __popc(s_bm[id >> 5] & ((1u << (id & 31)) - 1u));
                int sl = t;
                if (t >= O) sl = gg_reservoir_pick((unsigned long long)(long long)gi + 2ull * seed, t + 1);
                if (sl < O) atomicMax(&s_slot[sl], id + 1);
            }
            if (pop_[j] > P) {
                // S0: item n < P sits in slot n; item n >= P overwrites slot r(n) if r(n) < P
                // (gridify.cu:146-153).  Last writer = largest n = largest id.
                const int nn = pos - st_[j];
                int sl = nn;
                if (nn >= P) sl = gg_reservoir_pick((unsigned long long)(long long)gi + seed, nn + 1);
                if (sl < P) atomicMax(&s_bkt[st_[j] + sl], id);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < IPT; j++) {
        const int pos = tid * IPT + j;
        if (pos < n && pop_[j] > P && pos - st_[j] < P) a.bkt[gbase + pos] = s_bkt[pos];
    }
    for (int o = tid; o < O; o += 1024) a.slotfirst1[(size_t)b * O + o] = s_slot[o];
    if (tid == 0) {
        long long ws = 0;
        int ff = 0;
#pragma unroll
        for (int w = 0; w < GG_SM_NW; w++) { ws += s_sum[w]; ff |= s_flag[w]; }
        a.centnum[b] = nheads < O ? nheads : O;
        // integer weights with sum |w| < 2^23: S0's total_weight may be summed in any order (bit 0);
        // every in-grid weight exactly 1 (bit 1)
        const int ex = (!(ff & 1) && ws < (1ll << 23)) ? 1 : 0;
        a.exact[b] = ex | ((ex && !(ff & 2)) ? 2 : 0);
    }
}

// ------------------------------------------------------------------------------------------
static unsigned gg_inv_odd(unsigned a)  // inverse of an odd number mod 2^32 (Newton)
{
    unsigned x = a;
    for (int i = 0; i < 5; i++) x *= 2u - a * x;
    return x;
}

// plan overrides for measurements (include/gridgcn.h: gridgcn_set_option): shift of log2(slabs per
// cloud), points per chunk (0 = automatic)
static int g_opt_kb_shift = 0, g_opt_chunk = 0, g_opt_small = 1;
void gg_index_set_tuning(int which, int value)
{
    if (which == 0) g_opt_kb_shift = value;
    else if (which == 1) g_opt_chunk = value;
    else g_opt_small = value ? 1 : 0;
}
int gg_index_get_tuning(int which) { return which == 0 ? g_opt_kb_shift : (which == 1 ? g_opt_chunk : g_opt_small); }

static bool gg_small_ok(int N, const GGGrid &gp, bool with_centres)
{
    return g_opt_small && N <= GG_SM_MAXN && (!with_centres || gp.O <= GG_SM_MAXO);
}

static bool gg_plan(int B, int N, const GGGrid &gp, GGIndexWs *w)
{
    const long long nruns = ((long long)gp.G + (1 << GG_XRB) - 1) >> GG_XRB;
    int MB = 0;
    while ((1ll << MB) < nruns) MB++;
    // slabs: 2^KB per cloud.  At least so many that a slab has <= 4096 voxels; more while the
    // launch is below 1024 workgroups and a slab still gets >= 256 points on average, or while
    // a slab has more than 1024 voxels and still gets >= 128 points.
    int KB = MB - (GG_MAX_SB - GG_XRB);
    KB = KB < 0 ? 0 : KB;
    while (KB < MB && KB < 10) {
        const long long after = (long long)N >> (KB + 1);
        const bool grow = ((long long)B << KB) < 1024 && after >= 256;
        const bool shrink = (MB - KB + GG_XRB) > 10 && after >= 128;
        if (!grow && !shrink) break;
        KB++;
    }
    KB += g_opt_kb_shift;
    KB = KB < 0 ? 0 : (KB > MB ? MB : KB);
    if (KB > 10 || MB - KB + GG_XRB > GG_MAX_SB) return false;
    w->KB = KB;
    w->MB = MB;
    w->SB = MB - KB + GG_XRB;
    w->S = 1 << w->SB;
    w->nslab = 1 << KB;
    w->HA = 0x9E3779B1u;
    w->HAinv = gg_inv_odd(w->HA);
    w->NW2 = w->SB <= 9 ? 8 : (w->SB == 10 ? 4 : 2);  // per-wave counters <= 16 KB (32 KB at S = 4096)
    // chunks of 1024 / 2048 / 4096 points: small chunks while the launch stays within one wave
    // of workgroups (2 per CU), and never more than GG_MAX_CHUNKS per cloud
    int CH = 1024;
    while (CH < GG_CHUNK_MAX &&
           ((long long)B * ((N + CH - 1) / CH) > 512 || (N + CH - 1) / CH > GG_MAX_CHUNKS))
        CH *= 2;
    if ((g_opt_chunk == 1024 || g_opt_chunk == 2048 || g_opt_chunk == 4096) &&
        (N + g_opt_chunk - 1) / g_opt_chunk <= GG_MAX_CHUNKS)
        CH = g_opt_chunk;
    w->CH = CH;
    w->nblk = (N + CH - 1) / CH;
    if (w->nblk > GG_MAX_CHUNKS) return false;
    return true;
}

static GGSplit gg_split_of(const GGIndexWs &w)
{
    GGSplit sp;
    sp.HA = w.HA;
    sp.HAinv = w.HAinv;
    sp.mmask = w.MB >= 32 ? 0xffffffffu : ((1u << w.MB) - 1u);
    sp.LB = w.MB - w.KB;
    sp.SB = w.SB;
    sp.KB = w.KB;
    sp.nslab = w.nslab;
    sp.nchunk = w.nblk;
    sp.CH = w.CH;
    return sp;
}

static size_t gg_k1_lds(int nslab, int CH) { return (size_t)(GG_NW1 * nslab + CH) * 4; }
static size_t gg_k2_lds(int SB, int nchunk, int NW)
{
    const size_t S = (size_t)1 << SB;
    const size_t capw = NW >= 8 ? 256 : 512;
    return (NW * S + (S + 1) + S + (nchunk + 1) + nchunk + NW * capw) * 4 + NW * capw * 2;
}

size_t gg_index_workspace_bytes(int B, int N, const GGGrid &gp, bool with_centres, GGIndexWs *ws)
{
    GGIndexWs w = {};
    const size_t BG = (size_t)B * gp.G, BN = (size_t)B * N;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    if (gg_small_ok(N, gp, with_centres)) {
        // one-launch build: only what the query kernels read
        w.o_slotfirst1 = take(with_centres ? (size_t)B * gp.O * 4 : 0);
        w.o_vtab = take(BG * 8);
        w.o_sorted = take(BN * 4);
        w.o_bkt = take(with_centres ? BN * 4 : 0);
        w.o_exact = take((size_t)B * 4);
        w.total = o;
        w.small = 1;
        if (ws) *ws = w;
        return o;
    }
    if (!gg_plan(B, N, gp, &w))
        return gg_index_legacy_workspace_bytes(B, N, gp, with_centres, ws);
    // ---- zeroed by K1 ----
    w.o_slotfirst1 = take(with_centres ? (size_t)B * gp.O * 4 : 0);
    w.o_lbm = take(with_centres ? (size_t)B * ((N + 31) / 32) * 4 : 0);
    w.zero_bytes = o;
    // ---- written before read ----
    w.o_vtab = take(BG * 8);
    w.o_sorted = take(BN * 4);
    w.o_bkt = take(with_centres ? BN * 4 : 0);
    w.o_exact = take((size_t)B * 4);
    w.o_part = take(BN * 4);
    w.o_ctab = take((size_t)B * w.nblk * (w.nslab + 1) * 4);
    w.o_wsum = take((size_t)B * w.nblk * 8);
    w.total = o;
    w.legacy = 0;
    if (ws) *ws = w;
    return o;
}

template <bool WC, int NW>
static void gg_launch_k2(const GGSlabArgs &a, const GGGrid &gp, const GGSplit &sp, int B,
                         hipStream_t st)
{
    gg_k_slab_build<WC, NW><<<sp.nslab * B, 64 * NW, gg_k2_lds(sp.SB, sp.nchunk, NW), st>>>(
        a, B, gp, sp);
}

int gg_index_build(const float *data, const int *np, int B, int N, const GGGrid &gp,
                   bool with_centres, int *centnum, char *wsbase, const GGIndexWs &w,
                   hipStream_t st)
{
    if (w.legacy)
        return gg_index_legacy_build(data, np, B, N, gp, with_centres, centnum, wsbase, w, st);
    if (w.small) {
        GGSmallArgs a;
        a.data = (const float4 *)data;
        a.np = np;
        a.vtab = (int2 *)(wsbase + w.o_vtab);
        a.sorted = (int *)(wsbase + w.o_sorted);
        a.bkt = with_centres ? (int *)(wsbase + w.o_bkt) : nullptr;
        a.slotfirst1 = with_centres ? (int *)(wsbase + w.o_slotfirst1) : nullptr;
        a.centnum = centnum;
        a.exact = (int *)(wsbase + w.o_exact);
        a.N = N;
        a.B = B;
        a.nbits = 0;
        while (a.nbits < 24 && (1 << a.nbits) < gp.G) a.nbits++;
        const size_t lds = gg_small_lds(with_centres);
        const int ipt = N <= 1024 ? 1 : (N <= 2048 ? 2 : 4);
        if (with_centres) {
            if (ipt == 1) gg_k_small_build<1, true><<<B, 1024, lds, st>>>(a, gp);
            else if (ipt == 2) gg_k_small_build<2, true><<<B, 1024, lds, st>>>(a, gp);
            else gg_k_small_build<4, true><<<B, 1024, lds, st>>>(a, gp);
        } else {
            if (ipt == 1) gg_k_small_build<1, false><<<B, 1024, lds, st>>>(a, gp);
            else if (ipt == 2) gg_k_small_build<2, false><<<B, 1024, lds, st>>>(a, gp);
            else gg_k_small_build<4, false><<<B, 1024, lds, st>>>(a, gp);
        }
        return hipGetLastError() == hipSuccess ? 0 : 3;
    }
    const GGSplit sp = gg_split_of(w);
    unsigned *part = (unsigned *)(wsbase + w.o_part);
    int *ctab = (int *)(wsbase + w.o_ctab);
    unsigned long long *wsum = (unsigned long long *)(wsbase + w.o_wsum);
    const int g1 = sp.nchunk * B;
    const size_t l1 = gg_k1_lds(sp.nslab, sp.CH);
    const float4 *d4 = (const float4 *)data;
    int *zb = (int *)wsbase;
    const int zw = (int)(w.zero_bytes / 4);
    if (sp.CH == 1024)
        gg_k_chunk_split<1><<<g1, GG_NT1, l1, st>>>(d4, np, N, B, gp, sp, part, ctab, wsum, zb, zw);
    else if (sp.CH == 2048)
        gg_k_chunk_split<2><<<g1, GG_NT1, l1, st>>>(d4, np, N, B, gp, sp, part, ctab, wsum, zb, zw);
    else
        gg_k_chunk_split<4><<<g1, GG_NT1, l1, st>>>(d4, np, N, B, gp, sp, part, ctab, wsum, zb, zw);
    GGSlabArgs a;
    a.part = part;
    a.ctab = ctab;
    a.vtab = (int2 *)(wsbase + w.o_vtab);
    a.sorted = (int *)(wsbase + w.o_sorted);
    a.bkt = with_centres ? (int *)(wsbase + w.o_bkt) : nullptr;
    a.lbm = with_centres ? (unsigned *)(wsbase + w.o_lbm) : nullptr;
    a.N = N;
    if (with_centres) {
        if (w.NW2 == 8) gg_launch_k2<true, 8>(a, gp, sp, B, st);
        else if (w.NW2 == 4) gg_launch_k2<true, 4>(a, gp, sp, B, st);
        else gg_launch_k2<true, 2>(a, gp, sp, B, st);
        const int nw32 = (N + 31) / 32;
        gg_k_centre_slots<<<B * ((nw32 + GG_NT3 - 1) / GG_NT3), GG_NT3, 0, st>>>(
            N, B, gp, sp.nchunk, a.lbm, wsum, (int *)(wsbase + w.o_slotfirst1), centnum,
            (int *)(wsbase + w.o_exact));
    } else {
        if (w.NW2 == 8) gg_launch_k2<false, 8>(a, gp, sp, B, st);
        else if (w.NW2 == 4) gg_launch_k2<false, 4>(a, gp, sp, B, st);
        else gg_launch_k2<false, 2>(a, gp, sp, B, st);
    }
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_index_init()
{
    // dynamic LDS above the 64 KB default
    const int l1 = (int)gg_k1_lds(GG_MAX_SLABS, GG_CHUNK_MAX);
    const void *k1[3] = {(const void *)gg_k_chunk_split<1>, (const void *)gg_k_chunk_split<2>,
                         (const void *)gg_k_chunk_split<4>};
    for (int i = 0; i < 3; i++)
        if (hipFuncSetAttribute(k1[i], hipFuncAttributeMaxDynamicSharedMemorySize, l1) != hipSuccess)
            return 3;
    const void *k2[6] = {(const void *)gg_k_slab_build<true, 8>, (const void *)gg_k_slab_build<true, 4>,
                         (const void *)gg_k_slab_build<true, 2>, (const void *)gg_k_slab_build<false, 8>,
                         (const void *)gg_k_slab_build<false, 4>, (const void *)gg_k_slab_build<false, 2>};
    const int nw2[6] = {8, 4, 2, 8, 4, 2};
    const int sb2[6] = {9, 10, 12, 9, 10, 12};
    for (int i = 0; i < 6; i++)
        if (hipFuncSetAttribute(k2[i], hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)gg_k2_lds(sb2[i], GG_MAX_CHUNKS, nw2[i])) != hipSuccess)
            return 3;
    const void *k3[6] = {(const void *)gg_k_small_build<1, true>, (const void *)gg_k_small_build<2, true>,
                         (const void *)gg_k_small_build<4, true>, (const void *)gg_k_small_build<1, false>,
                         (const void *)gg_k_small_build<2, false>, (const void *)gg_k_small_build<4, false>};
    for (int i = 0; i < 6; i++)
        if (hipFuncSetAttribute(k3[i], hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)gg_small_lds(i < 3)) != hipSuccess)
            return 3;
    return gg_index_legacy_init();
}
