// gridgcn_index_legacy.hip -- first-generation voxel index build (gfx950), kept ONLY as the
// fallback for grids / clouds outside the domain of the two-level split in gridgcn_index.hip
// (more than 1024 slabs of 4096 voxels, or more than 1024 chunks of 4096 points per cloud) and
// for A/B runs (GG_INDEX_LEGACY=1).
//
// Replaces gridify_kernel_build_index (gridifyop/gridify.cu:102-191, gridifyknn.cu:115-204,
// gridify_up.cu:102-170).  The reference appends points to a dense [B*G, P] bucket table with
// atomics in arrival order (non-deterministic, 262-524 MB of scratch per call).  Here the result
// of the canonical schedule S0 (threads in ascending index) is computed order-independently:
//
//   K1 voxelize : voxel id per point (coalesced float4 stream, no atomics)
//   K2 slabs    : LDS-staged voxel slabs: per-voxel population + arrival slot via LDS atomics,
//                 LDS scan + one bump allocation per slab -> compact segment offsets
//   K3 scatter  : point ids into their voxel's segment in arrival order (order irrelevant)
//   K4 rank     : rank n of a point inside its voxel = #{ids in the segment smaller than mine};
//                 sorted[off+n] = id; reservoir of S0 resolved as "largest n wins" == atomicMax
//                 on the point id (ids ascend with n); voxel leaders (n == 0) flagged
//   K5 centres  : t = number of leaders before mine (block prefix + in-block scan) = the order of
//                 first appearance of the voxel; centre reservoir again "largest t wins"
//
// Every atomic used is commutative/idempotent on the final value (arrival slots and segment
// placement only permute scratch), so the output is bit-identical from run to run and equal to
// schedule S0 of the reference.
#include "gridgcn_index.h"

// ------------------------------------------------------------------------------------------
// K1: one thread per point, coalesced float4 loads, NO atomics.  grid (ceil(N/1024), B).
// Also reduces the per-block weight statistic that selects the exact-integer total_weight path.
__global__ __launch_bounds__(1024) void gg_k_voxelize(const float4 *__restrict__ data,
                                                      const int *__restrict__ np, int N, GGGrid gp,
                                                      int *__restrict__ vox,
                                                      unsigned long long *__restrict__ wsum_blk)
{
    __shared__ long long sw[16];
    __shared__ int sbad[16];
    const int b = blockIdx.y;
    const int ip = blockIdx.x * 1024 + threadIdx.x;
    const int nvalid = np[b];
    int v = -1;
    long long aw = 0;
    bool bad = false;
    if (ip < N && ip < nvalid) {
        float4 p = data[(size_t)b * N + ip];
        v = gg_voxel_of(p.x, p.y, p.z, gp, nullptr);
        if (v >= 0) {
            float w = p.w;
            bad = !(truncf(w) == w) || !(fabsf(w) < 8388608.0f);
            aw = bad ? 0 : (long long)fabsf(w);
        }
    }
    if (ip < N) vox[(size_t)b * N + ip] = v;
    long long s = gg_wave_sum_ll(aw);
    bool anybad = __any(bad);
    if (gg_lane() == 0) { sw[threadIdx.x >> 6] = s; sbad[threadIdx.x >> 6] = anybad ? 1 : 0; }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
        int nb = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) { t += sw[w]; nb |= sbad[w]; }
        // bit 63 = "some weight is not a small integer"
        wsum_blk[(size_t)b * gridDim.x + blockIdx.x] =
            (unsigned long long)t | (nb ? (1ull << 63) : 0ull);
    }
}

// ------------------------------------------------------------------------------------------
// K2: LDS-staged voxel slabs.  grid (nslab, B), block 1024, dynamic LDS = S ints.
// A workgroup owns the contiguous voxel range [s*S, (s+1)*S) of one cloud.  It streams the
// cloud's voxel ids (4 B per point, L2 resident) and counts its own voxels with LDS atomics:
// the returned value is the arrival slot of the point inside its voxel.  (Global returning
// atomics run memory-side on MI355X at only ~5 G/s -- measured 140 us for 655k points; LDS
// atomics make this kernel a pure L2 stream.)  Then: LDS exclusive scan of the slab's
// populations, one bump allocation per slab for its compact segment range, coalesced stores of
// cnt[] and off[] -- so no dense memset and no global scan pass either.
__global__ __launch_bounds__(1024) void gg_k_slab_count(const int *__restrict__ vox, int N, int G,
                                                        int S, int *__restrict__ arr,
                                                        int *__restrict__ cnt,
                                                        int *__restrict__ off,
                                                        int *__restrict__ cursor)
{
    extern __shared__ __attribute__((aligned(16))) int lcnt[];
    __shared__ int swc[16];
    __shared__ int sbase;
    const int b = blockIdx.y;
    const int v0 = blockIdx.x * S;
    const int v1 = (v0 + S < G) ? v0 + S : G;
    const int ns = v1 - v0;
    for (int j = threadIdx.x; j < ns; j += 1024) lcnt[j] = 0;
    __syncthreads();
    const int *vb = vox + (size_t)b * N;
    int *ab = arr + (size_t)b * N;
    const int N4 = ((((size_t)b * N) & 3) == 0) ? (N >> 2) : 0;  // int4 path needs 16 B alignment
    // four 16-byte loads in flight per thread (measured: no change -- the kernel's 20 us at
    // N = 81920 are the LDS atomics and the scattered arr[] stores, not the vox stream)
    for (int q0 = threadIdx.x; q0 < N4; q0 += 4 * 1024) {
        int4 v4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int q = q0 + u * 1024;
            v4[u] = q < N4 ? ((const int4 *)vb)[q] : make_int4(-1, -1, -1, -1);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int q = q0 + u * 1024;
            const int vv[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int v = vv[j];
                if (v >= v0 && v < v1) ab[q * 4 + j] = atomicAdd(&lcnt[v - v0], 1);
            }
        }
    }
    for (int i = N4 * 4 + threadIdx.x; i < N; i += 1024) {
        int v = vb[i];
        if (v >= v0 && v < v1) ab[i] = atomicAdd(&lcnt[v - v0], 1);
    }
    __syncthreads();
    // exclusive scan of lcnt[0..ns): each thread owns a contiguous run of `per` entries
    const int per = (ns + 1023) / 1024;
    const int j0 = threadIdx.x * per;
    int s = 0;
    for (int j = j0; j < j0 + per && j < ns; j++) s += lcnt[j];
    int incl = gg_wave_incl_scan(s);
    if (gg_lane() == 63) swc[threadIdx.x >> 6] = incl;
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
        int t = swc[w];
        if (w < (int)(threadIdx.x >> 6)) wbase += t;
        total += t;
    }
    if (threadIdx.x == 0) sbase = b * N + (total ? atomicAdd(&cursor[b], total) : 0);
    __syncthreads();
    int run = sbase + wbase + incl - s;
    size_t gb = (size_t)b * G + v0;
    for (int j = j0; j < j0 + per && j < ns; j++) {
        int c = lcnt[j];
        cnt[gb + j] = c;
        off[gb + j] = run;
        run += c;
    }
}

// ------------------------------------------------------------------------------------------
// K3: arrival-order scatter into the voxel segment; also resets the reservoir slots.
__global__ __launch_bounds__(256) void gg_k_scatter(int N, int G, const int *__restrict__ vox,
                                                    const int *__restrict__ arr,
                                                    const int *__restrict__ off,
                                                    int *__restrict__ seg, int *__restrict__ bkt)
{
    const int b = blockIdx.y;
    const int ip = blockIdx.x * 256 + threadIdx.x;
    if (ip >= N) return;
    size_t i = (size_t)b * N + ip;
    int v = vox[i];
    if (bkt) bkt[i] = -1;
    if (v >= 0) seg[off[(size_t)b * G + v] + arr[i]] = ip;
}

// ------------------------------------------------------------------------------------------
// K4: rank inside the voxel, sorted segment, bucket reservoir (gridify.cu:145-154), leaders.
// grid (ceil(N/1024), B), block 1024.  WITH_CENTRES=false for GridifyUp (no buckets/leaders).
template <bool WITH_CENTRES>
__global__ __launch_bounds__(1024) void gg_k_rank(int N, GGGrid gp, const int *__restrict__ vox,
                                                  const int *__restrict__ cnt,
                                                  const int *__restrict__ off,
                                                  const int *__restrict__ seg,
                                                  int *__restrict__ sorted, int *__restrict__ bkt,
                                                  unsigned char *__restrict__ lead,
                                                  int *__restrict__ blkcnt)
{
    __shared__ int swc[16];
    const int b = blockIdx.y;
    const int ip = blockIdx.x * 1024 + threadIdx.x;
    const size_t i = (size_t)b * N + ip;
    int v = (ip < N) ? vox[i] : -1;
    int is_lead = 0;
    if (v >= 0) {
        size_t vb = (size_t)b * gp.G + v;
        int c = cnt[vb];
        int o = off[vb];
        int n = 0;
        int j = 0;
        for (; j + 4 <= c; j += 4) {
            int a0 = seg[o + j], a1 = seg[o + j + 1], a2 = seg[o + j + 2], a3 = seg[o + j + 3];
            n += (a0 < ip) + (a1 < ip) + (a2 < ip) + (a3 < ip);
        }
        for (; j < c; j++) n += (seg[o + j] < ip);
        sorted[o + n] = ip;
        if (WITH_CENTRES) {
            if (c > gp.P) {
                // S0: item n < P sits in slot n; item n >= P overwrites slot r(n) if r(n) < P
                // (gridify.cu:146-153).  Last writer = largest n = largest point id.
                int s = n;
                if (n >= gp.P)
                    s = gg_reservoir_pick((unsigned long long)(long long)(int)i + gg_seed(gp), n + 1);
                if (s < gp.P) atomicMax(&bkt[o + s], ip);
            }
            is_lead = (n == 0);
        }
    }
    if (WITH_CENTRES) {
        if (ip < N) lead[i] = (unsigned char)is_lead;
        unsigned long long m = __ballot(is_lead);
        if (gg_lane() == 0) swc[threadIdx.x >> 6] = __popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
#pragma unroll
            for (int w = 0; w < 16; w++) t += swc[w];
            blkcnt[(size_t)b * gridDim.x + blockIdx.x] = t;
        }
    }
}

template __global__ void gg_k_rank<true>(int, GGGrid, const int *, const int *, const int *,
                                         const int *, int *, int *, unsigned char *, int *);
template __global__ void gg_k_rank<false>(int, GGGrid, const int *, const int *, const int *,
                                          const int *, int *, int *, unsigned char *, int *);

// ------------------------------------------------------------------------------------------
// K5: centre slots = RVS reservoir over voxels in order of first appearance (gridify.cu:165-189).
// slotfirst1[b,O] holds (first point id of the chosen voxel) + 1, 0 = empty.
__global__ __launch_bounds__(1024) void gg_k_centres(int N, GGGrid gp,
                                                     const unsigned char *__restrict__ lead,
                                                     const int *__restrict__ blkcnt,
                                                     const unsigned long long *__restrict__ wsum_blk,
                                                     int *__restrict__ slotfirst1,
                                                     int *__restrict__ centnum,
                                                     int *__restrict__ exact)
{
    __shared__ int swc[16];
    __shared__ int sred[16];
    const int b = blockIdx.y;
    const int nblk = gridDim.x;
    const int ip = blockIdx.x * 1024 + threadIdx.x;
    const size_t i = (size_t)b * N + ip;
    // leaders in the preceding blocks of this cloud (and, for block 0, in the whole cloud)
    int before = 0, all = 0;
    for (int j = threadIdx.x; j < nblk; j += 1024) {
        int c = blkcnt[(size_t)b * nblk + j];
        all += c;
        if (j < (int)blockIdx.x) before += c;
    }
    before = gg_wave_sum(before);
    all = gg_wave_sum(all);
    if (gg_lane() == 0) { swc[threadIdx.x >> 6] = before; sred[threadIdx.x >> 6] = all; }
    __syncthreads();
    int t0 = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) { t0 += swc[w]; total += sred[w]; }
    __syncthreads();
    int flag = (ip < N) ? (int)lead[i] : 0;
    unsigned long long m = __ballot(flag);
    int pre = __popcll(m & ((1ull << gg_lane()) - 1ull));
    if (gg_lane() == 0) swc[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) wbase += swc[w];
    if (flag) {
        int t = t0 + wbase + pre;
        int s = t;
        if (t >= gp.O)
            s = gg_reservoir_pick((unsigned long long)(long long)(int)i + 2ull * gg_seed(gp), t + 1);
        if (s < gp.O) atomicMax(&slotfirst1[(size_t)b * gp.O + s], ip + 1);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        centnum[b] = total < gp.O ? total : gp.O;
        // weights of the cloud are integers and sum(|w|) < 2^23: every partial sum of S0's
        // total_weight accumulation is exact, so it may be evaluated in any order
        unsigned long long ws = 0, bad = 0;
        for (int j = 0; j < nblk; j++) {
            unsigned long long x = wsum_blk[(size_t)b * nblk + j];
            bad |= x >> 63;
            ws += x & ~(1ull << 63);
        }
        exact[b] = (!bad && ws < (1ull << 23)) ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------
// the query kernels read one int2 per voxel
__global__ __launch_bounds__(256) void gg_k_legacy_pack_vtab(const int *__restrict__ cnt,
                                                             const int *__restrict__ off,
                                                             int2 *__restrict__ vtab, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) vtab[i] = make_int2(off[i], cnt[i]);
}

size_t gg_index_legacy_workspace_bytes(int B, int N, const GGGrid &gp, bool with_centres, GGIndexWs *ws)
{
    const size_t BG = (size_t)B * gp.G, BN = (size_t)B * N;
    const int nblk = (N + 1023) / 1024;
    // slabs: enough workgroups to fill 256 CUs twice, each slab <= 32768 voxels (128 KB of LDS)
    int nslab = (512 + B - 1) / B;
    const int min_slab = (gp.G + 32767) / 32768;
    if (nslab < min_slab) nslab = min_slab;
    if (nslab > gp.G) nslab = gp.G;
    const int S = (gp.G + nslab - 1) / nslab;
    nslab = (gp.G + S - 1) / S;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    GGIndexWs w = {};
    w.legacy = 1;
    // ---- zero-filled region (one hipMemsetAsync) ----
    w.o_slotfirst1 = take(with_centres ? (size_t)B * gp.O * 4 : 0);
    w.o_cursor = take((size_t)B * 4);
    w.zero_bytes = o;
    // ---- written before read ----
    w.o_cnt = take(BG * 4);
    w.o_off = take(BG * 4);
    w.o_vtab = take(BG * 8);
    w.o_blkcnt = take(with_centres ? (size_t)B * nblk * 4 : 0);
    w.o_wsum = take((size_t)B * nblk * 8);
    w.o_exact = take((size_t)B * 4);
    w.o_vox = take(BN * 4);
    w.o_arr = take(BN * 4);
    w.o_seg = take(BN * 4);
    w.o_sorted = take(BN * 4);
    w.o_bkt = take(with_centres ? BN * 4 : 0);
    w.o_leadflag = take(with_centres ? BN : 0);
    w.total = o;
    w.nblk = nblk;
    w.nslab = nslab;
    w.S = S;
    if (ws) *ws = w;
    return o;
}

int gg_index_legacy_build(const float *data, const int *np, int B, int N, const GGGrid &gp,
                   bool with_centres, int *centnum, char *wsbase, const GGIndexWs &w,
                   hipStream_t st)
{
    int *cnt = (int *)(wsbase + w.o_cnt), *off = (int *)(wsbase + w.o_off);
    int *vox = (int *)(wsbase + w.o_vox);
    int *arr = (int *)(wsbase + w.o_arr), *seg = (int *)(wsbase + w.o_seg);
    int *sorted = (int *)(wsbase + w.o_sorted);
    int *bkt = with_centres ? (int *)(wsbase + w.o_bkt) : nullptr;
    unsigned char *lead = with_centres ? (unsigned char *)(wsbase + w.o_leadflag) : nullptr;
    int *slotfirst1 = (int *)(wsbase + w.o_slotfirst1), *blkcnt = (int *)(wsbase + w.o_blkcnt);
    unsigned long long *wsum = (unsigned long long *)(wsbase + w.o_wsum);
    int *exact = (int *)(wsbase + w.o_exact);
    int *cursor = (int *)(wsbase + w.o_cursor);

    if (hipMemsetAsync(wsbase, 0, w.zero_bytes, st) != hipSuccess) return 3;
    dim3 g256((N + 255) / 256, B), g1024(w.nblk, B), gslab(w.nslab, B);
    gg_k_voxelize<<<g1024, 1024, 0, st>>>((const float4 *)data, np, N, gp, vox, wsum);
    gg_k_slab_count<<<gslab, 1024, (size_t)w.S * 4, st>>>(vox, N, gp.G, w.S, arr, cnt, off,
                                                          cursor);
    gg_k_scatter<<<g256, 256, 0, st>>>(N, gp.G, vox, arr, off, seg, bkt);
    if (with_centres) {
        gg_k_rank<true><<<g1024, 1024, 0, st>>>(N, gp, vox, cnt, off, seg, sorted, bkt, lead,
                                                blkcnt);
        gg_k_centres<<<g1024, 1024, 0, st>>>(N, gp, lead, blkcnt, wsum, slotfirst1, centnum,
                                             exact);
    } else {
        gg_k_rank<false><<<g1024, 1024, 0, st>>>(N, gp, vox, cnt, off, seg, sorted, nullptr,
                                                 nullptr, nullptr);
    }
    {
        const size_t n = (size_t)B * gp.G;
        gg_k_legacy_pack_vtab<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(
            cnt, off, (int2 *)(wsbase + w.o_vtab), n);
    }
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_index_legacy_init() {
    // slabs may use up to 128 KB of dynamic LDS (default limit is 64 KB)
    return hipFuncSetAttribute((const void *)gg_k_slab_count,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4) == hipSuccess
               ? 0 : 3;
}
