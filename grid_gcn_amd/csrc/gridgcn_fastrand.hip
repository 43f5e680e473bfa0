// gridgcn_fastrand.hip -- the `fast_rand` build variant of Gridify (gridifyop/fast_rand/gridify.cu),
// gfx950.  Same operator signature and outputs as Gridify, different sampling:
//   * build (:126-200): every point is scattered into the buckets of ALL k^3 voxels around its own
//     (one thread per (point, offset)); a bucket past P is a reservoir seeded with the THREAD index;
//     the centre voxels are the first max_o_grid occupied voxels in order of first appearance (no
//     reservoir over the centres);
//   * query (:232-272): a centre reads the bucket of its OWN voxel only (first min(count, P)
//     entries), pads with the first entry; loc == 0: centre = weighted mean of the picked points,
//     loc == 1: weighted mean of the voxel's own points.
// Under the canonical schedule S0 (threads in ascending index) the bucket of voxel V is the list of
// the points whose voxel lies in V's window, in ascending point id, with "the last writer of a slot
// wins" past P.  Nothing is scattered here: the shared sorted voxel index (gridgcn_index.hip) holds
// every voxel's points in ascending id, so a wave that owns a centre merges the <= k^3 sorted runs
// of its window on the fly -- rank of a candidate = its position in its own run + the lower bounds
// of its id in the other runs (runs staged in LDS when they fit) -- and plays the reservoir with
// an LDS atomicMax on the rank.
#include "gridgcn_index.h"
#include "gridgcn_once.h"

#define GG_FR_NT 1024
#define GG_FR_QW 4          // waves per workgroup of the query
#define GG_FR_CAP 1536      // candidate ids a wave stages in LDS

// ---- centre slots: the first O occupied voxels by first appearance (:165-186 under S0) --------
// one workgroup per cloud; first[] = first point per voxel (global scratch), leaders compacted in
// ascending id through a bitmap + block scan.
__global__ __launch_bounds__(GG_FR_NT) void gg_k_fastrand_slots(const float4 *__restrict__ data,
                                                                const int *__restrict__ np, int N,
                                                                GGGrid gp, int *__restrict__ first_g,
                                                                unsigned *__restrict__ bm_g, int W,
                                                                int *__restrict__ slotfirst1,
                                                                int *__restrict__ centnum)
{
    __shared__ int s_w[GG_FR_NT / 64];
    __shared__ int s_carry;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = gp.G, O = gp.O;
    int npts = np[b];
    npts = npts < 0 ? 0 : (npts > N ? N : npts);
    const float4 *cloud = data + (size_t)b * N;
    int *first = first_g + (size_t)b * G;
    unsigned *lbm = bm_g + (size_t)b * 2 * W;
    for (int v = tid; v < G; v += GG_FR_NT) first[v] = 0x7fffffff;
    for (int w = tid; w < W; w += GG_FR_NT) lbm[w] = 0u;
    for (int s = tid; s < O; s += GG_FR_NT) slotfirst1[(size_t)b * O + s] = 0;
    __syncthreads();
    for (int i = tid; i < npts; i += GG_FR_NT) {
        const float4 p = cloud[i];
        const int v = gg_voxel_of(p.x, p.y, p.z, gp, nullptr);
        if (v >= 0) atomicMin(&first[v], i);
    }
    __syncthreads();
    for (int v = tid; v < G; v += GG_FR_NT) {
        const int f = first[v];
        if (f != 0x7fffffff) atomicOr(&lbm[f >> 5], 1u << (f & 31));
    }
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int w0 = 0; w0 < W; w0 += GG_FR_NT) {
        const int w = w0 + tid;
        unsigned bits = w < W ? lbm[w] : 0u;
        const int cnt = __popc(bits);
        const int incl = gg_wave_incl_scan(cnt);
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        int base = s_carry;
        for (int x = 0; x < wave; x++) base += s_w[x];
        int t = base + incl - cnt;
        while (bits && t < O) {
            const int bit = __ffs(bits) - 1;
            bits &= bits - 1;
            slotfirst1[(size_t)b * O + t] = w * 32 + bit + 1;
            t++;
        }
        __syncthreads();
        if (tid == GG_FR_NT - 1) s_carry = base + incl;
        __syncthreads();
    }
    if (tid == 0) centnum[b] = s_carry < O ? s_carry : O;
}

// ---- query ------------------------------------------------------------------------------------
struct GGFastQ {
    const int2 *vtab;
    const int *sorted, *slotfirst1, *centnum;
};

__global__ __launch_bounds__(64 * GG_FR_QW) void gg_k_query_fastrand(
    const float4 *__restrict__ data, int N, GGGrid gp, GGFastQ q, int B, int *__restrict__ nebidx,
    float *__restrict__ nebmsk, float4 *__restrict__ cent, float *__restrict__ centmsk)
{
    extern __shared__ __attribute__((aligned(16))) int frl[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int P = gp.P, k = gp.k, k3 = gp.k3, hk = (k - 1) / 2;
    int *wl = frl + wave * (2 * k3 + 1 + 2 * P + GG_FR_CAP);
    int *s_start = wl;                 // [k3]
    int *s_excl = s_start + k3;        // [k3+1]
    int *s_bestc = s_excl + k3 + 1;    // [P] rank of the final occupant of a bucket slot
    int *s_id = s_bestc + P;           // [P]
    int *s_cand = s_id + P;            // [CAP] staged runs
    const long long index = (long long)blockIdx.x * GG_FR_QW + wave;   // centre slot, flat
    if (index >= (long long)B * gp.O) return;
    const int b = (int)(index / gp.O), o = (int)(index - (long long)b * gp.O);
    int *row = nebidx + index * P;
    float *mrow = nebmsk + index * P;
    const int cn = q.centnum[b];
    if (o >= cn) {                      // GridifyOp::Forward fill values (gridify-inl.h:117-121)
        for (int j = lane; j < P; j += 64) { row[j] = 0; mrow[j] = 0.f; }
        if (lane == 0) { cent[index] = make_float4(1.f, 1.f, 1.f, 1.f); centmsk[index] = 0.f; }
        return;
    }
    const float4 *cloud = data + (size_t)b * N;
    int c3[3];
    {
        const float4 p0 = cloud[q.slotfirst1[index] - 1];
        (void)gg_voxel_of(p0.x, p0.y, p0.z, gp, c3);
    }
    // neighbour table in (z,y,x) order, exclusive offsets
    int carry = 0;
    for (int l0 = 0; l0 < k3; l0 += 64) {
        const int l = l0 + lane;
        int2 vt = make_int2(0, 0);
        if (l < k3) {
            const int d = l / (k * k) - hk + c3[2], h = (l % (k * k)) / k - hk + c3[1],
                      w = l % k - hk + c3[0];
            if (d >= 0 && d < gp.g[2] && h >= 0 && h < gp.g[1] && w >= 0 && w < gp.g[0])
                vt = q.vtab[(size_t)b * gp.G + (size_t)d * gp.gxy + h * gp.g[0] + w];
        }
        const int incl = gg_wave_incl_scan(vt.y);
        if (l < k3) { s_start[l] = vt.x; s_excl[l] = carry + incl - vt.y; }
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) s_excl[k3] = carry;
    const int n = carry;                               // population of the window (>= 1)
    const int m = n < P ? n : P;
    for (int j = lane; j < P; j += 64) s_bestc[j] = j < m ? j : -1;
    __builtin_amdgcn_wave_barrier();
    const bool staged = n <= GG_FR_CAP;
    auto locate = [&](int g, int &l, int &t) {
        int lo = 0, hi = k3;                            // s_excl[lo] <= g < s_excl[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_excl[mid] <= g) lo = mid; else hi = mid;
        }
        l = lo;
        t = g - s_excl[lo];
    };
    if (staged) {
        for (int g = lane; g < n; g += 64) {
            int l, t;
            locate(g, l, t);
            s_cand[g] = q.sorted[s_start[l] + t];
        }
        __builtin_amdgcn_wave_barrier();
    }
    auto cand = [&](int l, int t) -> int {
        return staged ? s_cand[s_excl[l] + t] : q.sorted[s_start[l] + t];
    };
    // merged rank of candidate (l, t) with id: t + sum over the other runs of #ids below it
    auto rank_of = [&](int l, int t, int id) -> int {
        int c = t;
        for (int u = 0; u < k3; u++) {
            const int len = s_excl[u + 1] - s_excl[u];
            if (u == l || len == 0) continue;
            int lo = 0, hi = len;                       // first position with value > id
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (cand(u, mid) < id) lo = mid + 1; else hi = mid;
            }
            c += lo;
        }
        return c;
    };
    // slot a late candidate (rank c >= P) draws (:149-153; curand_init(threadindex)):
    //   threadindex = (b*N + id)*k^3 + j,  (threadindex + k^3/2) % k^3 = offset from ITS voxel to V
    auto draw = [&](int l, int id, int c) -> int {
        const int nei = k3 - 1 - l;                    // V seen from the candidate's voxel
        const int j = (nei - k3 / 2 + k3) % k3;
        const int tix = (int)(((long long)b * N + id) * k3 + j);
        return gg_reservoir_pick((unsigned long long)(long long)tix, c + 1);
    };
    if (n > P) {
        for (int g = lane; g < n; g += 64) {
            int l, t;
            locate(g, l, t);
            const int id = cand(l, t);
            const int c = rank_of(l, t, id);
            if (c >= P) {
                const int r = draw(l, id, c);
                if (r < P) atomicMax(&s_bestc[r], c);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    for (int g = lane; g < n; g += 64) {
        int l, t;
        locate(g, l, t);
        const int id = cand(l, t);
        const int c = (n > P || k3 > 1) ? rank_of(l, t, id) : t;
        if (c < P) {
            if (s_bestc[c] == c) s_id[c] = id;
        } else {
            const int r = draw(l, id, c);
            if (r < P && s_bestc[r] == c) s_id[r] = id;
        }
    }
    __builtin_amdgcn_wave_barrier();
    // outputs (:250-266)
    const int init = s_id[0];
    int wsum = 0;
    for (int j = lane; j < P; j += 64) {
        const int id = j < m ? s_id[j] : init;
        row[j] = id;
        mrow[j] = j < m ? 1.f : 0.f;
        if (j < m) wsum += (int)cloud[id].w;            // int in_data_eleweight (:255)
    }
    wsum = gg_wave_sum(wsum);
    if (lane == 0) {
        float xs = 0.f, ys = 0.f, zs = 0.f, cw = 0.f;
        if (gp.loc == 1) {
            // own voxel's points in ascending id = S0 order of the atomicAdds (:168-174)
            const int lc = (k3 - 1) / 2;
            const int len = s_excl[lc + 1] - s_excl[lc];
            for (int t = 0; t < len; t++) {
                const float4 p = cloud[cand(lc, t)];
                xs = __fadd_rn(xs, __fmul_rn(p.x, p.w));
                ys = __fadd_rn(ys, __fmul_rn(p.y, p.w));
                zs = __fadd_rn(zs, __fmul_rn(p.z, p.w));
                cw = __fadd_rn(cw, p.w);
            }
        } else {
            for (int j = 0; j < m; j++) {               // (:256-261)
                const float4 p = cloud[s_id[j]];
                const float ew = (float)(int)p.w;
                xs = __fadd_rn(xs, __fmul_rn(p.x, ew));
                ys = __fadd_rn(ys, __fmul_rn(p.y, ew));
                zs = __fadd_rn(zs, __fmul_rn(p.z, ew));
                cw = __fadd_rn(cw, ew);
            }
        }
        cent[index] = make_float4(__fdiv_rn(xs, cw), __fdiv_rn(ys, cw), __fdiv_rn(zs, cw),
                                  (float)wsum);
        centmsk[index] = 1.f;
    }
}

int gg_fastrand_query(const float *data, const int *np, int B, int N, const GGGrid &gp, char *wsbase,
                      const GGIndexWs &w, char *scratch, int *nebidx, float *nebmsk, float *cent,
                      float *centmsk, int *centnum, hipStream_t st)
{
    static GGDevOnce attr_done;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)gg_k_query_fastrand, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)
            return 3;
        attr_done = true;
    }
    const int W = (N + 31) / 32;
    int *first = (int *)scratch;
    unsigned *bm = (unsigned *)(scratch + (((size_t)B * gp.G * 4 + 255) & ~(size_t)255));
    int *slotfirst1 = (int *)(wsbase + w.o_slotfirst1);
    gg_k_fastrand_slots<<<B, GG_FR_NT, 0, st>>>((const float4 *)data, np, N, gp, first, bm, W,
                                                 slotfirst1, centnum);
    if (hipGetLastError() != hipSuccess) return 3;
    GGFastQ q;
    q.vtab = (const int2 *)(wsbase + w.o_vtab);
    q.sorted = (const int *)(wsbase + w.o_sorted);
    q.slotfirst1 = slotfirst1;
    q.centnum = centnum;
    const long long ncent = (long long)B * gp.O;
    const size_t lds = (size_t)GG_FR_QW * (2 * gp.k3 + 1 + 2 * gp.P + GG_FR_CAP) * sizeof(int);
    gg_k_query_fastrand<<<(unsigned)((ncent + GG_FR_QW - 1) / GG_FR_QW), 64 * GG_FR_QW, lds, st>>>(
        (const float4 *)data, N, gp, q, B, nebidx, nebmsk, (float4 *)cent, centmsk);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
