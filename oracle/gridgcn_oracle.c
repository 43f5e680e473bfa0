/*
 * gridgcn_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar, sequential) of the Grid-GCN index operators
 * under the canonical schedule S0: "the threads of each reference CUDA kernel run
 * one after another in ascending global thread index; kernels in launch order".
 * S0 is a legal schedule of the reference kernels, hence its output is a valid
 * reference output.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product path (grid_gcn_amd/) never links or calls it.
 *
 * PARITY PINNING: the reference ships no tests, no golden vectors and no CPU path
 * for Gridify/GridifyUp/GridifyKNN (gridifyop/gridify.cc:28-39 is LOG(FATAL)),
 * and neither MXNet nor nvcc exist in the build image, so this oracle cannot be
 * checked against reference outputs: "parity unpinned" by reference vectors.
 * It is pinned instead by (a) line-by-line citations below, (b) the
 * schedule-independent invariants of SURVEY App. A.7 (tests/test_oracle_invariants.py),
 * and (c) a second, independent numpy restatement for small cases
 * (tests/pyref.py).
 *
 * Third-party arithmetic restated here: cuRAND XORWOW curand_init(seed,0,0) +
 * one curand_uniform() (CUDA toolkit curand_kernel.h, version unpinned by the
 * reference Makefile).  Constants are the published ones; they cannot be verified
 * against a CUDA toolkit in this image.
 *
 * Floating point: fp32, IEEE, no contraction (build with -ffp-contract=off).
 *
 * Defined behaviour where the reference is undefined (documented in DESIGN.md):
 *   - a coordinate whose floor() is NaN / outside int range drops the point;
 *   - GridifyUp with an empty query voxel writes index 0 (reference: uninitialised
 *     initID, gridify_up.cu:211-220);
 *   - GridifyKNN slots beyond the number of candidates hold besti[0] and their
 *     weight is that of besti[0] (reference: uninitialised besti[], gridifyknn.cu:257-312);
 *   - KNN with fewer than k known points pads with -1 (reference: uninitialised
 *     besti[], k_nn-inl.h:61).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define DATA_NDIM 4

/* ---- cuRAND XORWOW, curand_init(seed, 0, 0) then first curand_uniform ------- */
/* call sites: gridify.cu:149-150, 182-183, 260-261; gridify_up.cu:162-163        */
/* The generator with its four seed-scramble constants as parameters: n-th raw 32-bit output (n >= 1)
 * of the state seeded with (seed, subsequence 0, offset 0).  Marsaglia's xorwow (state words, shifts 2 / 1 / 4,
 * Weyl increment 362437) + the seeding pattern x0 += t0, x1 ^= t0, x2 += t1, x3 ^= t1, x4 += t0, d += t1 + t0.
 * tests/test_oracle.py::test_xorwow_skeleton_is_rocrands runs this skeleton with rocRAND's constants against
 * rocRAND's own host-callable engine (/opt/rocm/include/rocrand/rocrand_xorwow.h, in the image): everything
 * but cuRAND's four scramble constants and the 2^-33 offset of _curand_uniform is checked there against a
 * vendor implementation of the same published generator. */
uint32_t gridgcn_oracle_xorwow_raw(uint64_t seed, uint32_t xs0, uint32_t xs1, uint32_t m0, uint32_t m1, int n)
{
    uint32_t s0 = ((uint32_t)seed) ^ xs0;
    uint32_t s1 = ((uint32_t)(seed >> 32)) ^ xs1;
    uint32_t t0 = m0 * s0;
    uint32_t t1 = m1 * s1;
    uint32_t d = 6615241u + t1 + t0;
    uint32_t v0 = 123456789u + t0;
    uint32_t v1 = 362436069u ^ t0;
    uint32_t v2 = 521288629u + t1;
    uint32_t v3 = 88675123u ^ t1;
    uint32_t v4 = 5783321u + t0;
    uint32_t x = 0;
    for (int i = 0; i < n; i++) {   /* one XORWOW step */
        uint32_t t = v0 ^ (v0 >> 2);
        v0 = v1; v1 = v2; v2 = v3; v3 = v4;
        v4 = (v4 ^ (v4 << 4)) ^ (t ^ (t << 1));
        d += 362437u;
        x = v4 + d;
    }
    return x;
}

float gridgcn_oracle_xorwow_uniform(uint64_t seed)
{
    /* cuRAND's scramble constants (curand_kernel.h, _curand_init_scratch): recalled, unverifiable here */
    uint32_t x = gridgcn_oracle_xorwow_raw(seed, 0xaad26b49u, 0xf7dcefddu, 1099087573u, 2591861531u, 1);
    /* _curand_uniform: x * 2^-32 + 2^-33, one rounding (product is exact) */
    return (float)x * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}

/* "int insrtidx = ceilf(curand_uniform(&state) * (n)) - 1;"  gridify.cu:150 */
/* OpenMP team size of the loops over clouds / queries: min(work items, gridgcn_oracle_set_threads()
 * or the OpenMP default).  A clause per region, so the calling process's own OpenMP settings
 * (PyTorch-CPU shares the runtime) are left alone. */
#ifdef _OPENMP
#include <omp.h>
#endif
static int g_oracle_threads = 0;
void gridgcn_oracle_set_threads(int n) { g_oracle_threads = n; }
static int oracle_threads(long long work)
{
    int t = g_oracle_threads;
#ifdef _OPENMP
    if (t <= 0) t = omp_get_max_threads();
#else
    t = 1;
#endif
    if ((long long)t > work) t = (int)work;
    return t < 1 ? 1 : t;
}
int gridgcn_oracle_threads(long long work) { return oracle_threads(work); }

static inline int reservoir_pick(uint64_t seed, int n)
{
    float u = gridgcn_oracle_xorwow_uniform(seed);
    return (int)(ceilf(u * (float)n) - 1.0f);
}

/* voxel of a point, gridify.cu:134-143.  returns -1 when the point is dropped. */
static inline int voxel_of(const float *p, const float *shift, const float *vs,
                           const int *grid, int *c_out)
{
    int c[3];
    for (int j = 0; j < 3; j++) {
        float q = (p[j] + shift[j]) / vs[j];
        float f = floorf(q);
        if (!(f >= 0.0f) || !(f < (float)grid[j])) return -1;
        c[j] = (int)f;
    }
    if (c_out) { c_out[0] = c[0]; c_out[1] = c[1]; c_out[2] = c[2]; }
    return c[2] * (grid[0] * grid[1]) + c[1] * grid[0] + c[0];
}

typedef struct {
    int *order_vox;  /* optional: voxel of first-appearance rank t        [G] (CAS only) */
    int *order_idx;  /* optional: global index of that voxel's first point [G] (CAS only) */
    int *cnt;        /* coor_counter         [G]   */
    int *bucket;     /* coor_to_pntidx       [G*P] */
    float *sums;     /* coor_to_locxyzw      [G*4] */
    unsigned char *touched; /* coor_to_voxelidx != -1 [G] */
    int *slot2vox;   /* voxelidx_to_coor     [O]   */
} build_tables;

/* gridify_kernel_build_index under S0 for one cloud, gridify.cu:126-190.
 * identical to gridifyKNN_kernel_build_index (gridifyknn.cu:115-204). */
static int build_index_cloud(const float *data, int b, int N, int np, int P, int O, int loc,
                             const float *shift, const float *vs, const int *grid,
                             uint64_t seed, build_tables *T, float *centmsk /* [O] */)
{
    int centcount = 0;
    for (int i = 0; i < N; i++) {
        if (!(i < np)) continue;                               /* :130 */
        int index = b * N + i;                                 /* :126-129 */
        const float *p = data + (size_t)index * DATA_NDIM;
        int v = voxel_of(p, shift, vs, grid, NULL);            /* :134-143 */
        if (v < 0) continue;
        int c = T->cnt[v]++;                                   /* :145 */
        if (c < P) {
            T->bucket[(size_t)v * P + c] = i;                  /* :146-147 */
        } else {
            int r = reservoir_pick((uint64_t)(int64_t)index + seed, c + 1); /* :149-150 */
            if (r < P) T->bucket[(size_t)v * P + r] = i;       /* :151-153 */
        }
        if (loc == 1) {                                        /* :155-162 */
            float w = p[3];
            T->sums[v * 4 + 0] += p[0] * w;
            T->sums[v * 4 + 1] += p[1] * w;
            T->sums[v * 4 + 2] += p[2] * w;
            T->sums[v * 4 + 3] += w;
        }
        if (!T->touched[v]) {                                  /* :165-172 */
            T->touched[v] = 1;
            int t = centcount++;                               /* :176 */
            if (T->order_vox) { T->order_vox[t] = v; T->order_idx[t] = index; }
            if (t < O) {
                T->slot2vox[t] = v;                            /* :179 */
                centmsk[t] = 1.0f;                             /* :180 */
            } else {
                int r = reservoir_pick((uint64_t)(int64_t)index + 2 * seed, t + 1); /* :182-183 */
                if (r < O) T->slot2vox[r] = v;                 /* :184-186 */
            }
        }
    }
    return centcount;
}

static void alloc_tables(build_tables *T, int G, int P, int O)
{
    T->order_vox = NULL;
    T->order_idx = NULL;
    T->cnt = (int *)calloc((size_t)G, sizeof(int));
    T->bucket = (int *)malloc((size_t)G * P * sizeof(int));
    T->sums = (float *)calloc((size_t)G * 4, sizeof(float));
    T->touched = (unsigned char *)calloc((size_t)G, 1);
    T->slot2vox = (int *)calloc((size_t)(O > 0 ? O : 1), sizeof(int));
}
static void free_tables(build_tables *T)
{
    free(T->order_vox); free(T->order_idx);
    free(T->cnt); free(T->bucket); free(T->sums); free(T->touched); free(T->slot2vox);
}

static void init_outputs_cloud(int O, int P, int *nebidx, float *nebmsk, float *cent,
                               float *centmsk, int *centnum)
{
    /* GridifyOp::Forward output fill, gridify-inl.h:117-121 */
    memset(nebidx, 0, (size_t)O * P * sizeof(int));
    memset(nebmsk, 0, (size_t)O * P * sizeof(float));
    for (int j = 0; j < O * 4; j++) cent[j] = 1.0f;
    memset(centmsk, 0, (size_t)O * sizeof(float));
    *centnum = 0;
}

/* ------------------------------ Gridify -------------------------------------- */
/* data[B,N,4] f32, actual_numpoints[B] i32 -> nebidx[B,O,P] i32, nebidxmsk[B,O,P] f32,
 * cent[B,O,4] f32, centmsk[B,O] f32, actual_centnum[B] i32   (gridify-inl.h:190-195) */
/* ---- Coverage-Aware Sampling (CAS) of the centre voxels: OUR specification ---------------------
 * PARITY UNPINNED.  The reference has no source for it (gridifyop/additional.so: Gridify_occaware*,
 * SURVEY F3); this is the greedy algorithm of the paper (Grid-GCN, CVPR 2020, section 3.2):
 *   incumbents  = the RVS sample of the build above: slots 0..M-1, M = min(#occupied, O)
 *   challengers = the occupied voxels that are NOT incumbents after the build, in order of first
 *                 appearance (rank t of gridify.cu:176); a voxel that loses its slot later is not
 *                 re-entered
 *   a challenger Vc meets ONE random incumbent: slot s = ceil(u*M) - 1, u = XORWOW(index + 3*seed),
 *                 index = global index of Vc's first point (the seed pattern of gridify.cu:182)
 *   H_add = sum_{V in win(Vc), occupied} [C_V == 0] - beta * C_V / lambda            (eq. 3)
 *   H_rmv = sum_{V in win(Vi), occupied} [C_V == 1]                                  (eq. 4)
 *   C_V = number of incumbents whose k^3 window (the query's window, gridify.cu:240-246) holds V,
 *   lambda = k^3.  H_add > H_rmv  <=>  lambda*(n0 - n1) > beta * sum C_V  (integer sums, one fp32
 *   multiply): Vc takes slot s, C is decremented over win(Vi) and incremented over win(Vc). */
static int cas_neighbour(int v, int nei, int ksz, const int *grid)
{
    const int gxy = grid[0] * grid[1];
    int c2 = v / gxy, c1 = (v - c2 * gxy) / grid[0], c0 = v - c2 * gxy - c1 * grid[0];
    int d = nei / (ksz * ksz) - (ksz - 1) / 2 + c2;
    int h = (nei % (ksz * ksz)) / ksz - (ksz - 1) / 2 + c1;
    int w = nei % ksz - (ksz - 1) / 2 + c0;
    if (d < 0 || d >= grid[2] || h < 0 || h >= grid[1] || w < 0 || w >= grid[0]) return -1;
    return d * gxy + h * grid[0] + w;
}

static void cas_refine_cloud(build_tables *T, int centcount, int O, int ksz, const int *grid,
                             uint64_t seed, float beta)
{
    if (centcount <= O) return;                    /* every occupied voxel already is a centre */
    const int G = grid[0] * grid[1] * grid[2], size = ksz * ksz * ksz, M = O;
    int *cov = (int *)calloc((size_t)G, sizeof(int));
    unsigned char *picked = (unsigned char *)calloc((size_t)G, 1);
    for (int s = 0; s < M; s++) {
        int v = T->slot2vox[s];
        picked[v] = 1;
        for (int nei = 0; nei < size; nei++) {
            int u = cas_neighbour(v, nei, ksz, grid);
            if (u >= 0 && T->touched[u]) cov[u]++;
        }
    }
    for (int t = 0; t < centcount; t++) {
        int vc = T->order_vox[t];
        if (picked[vc]) continue;                  /* incumbent of the initial sample */
        int s = reservoir_pick((uint64_t)(int64_t)T->order_idx[t] + 3 * seed, M);
        int vi = T->slot2vox[s];
        int n0 = 0, n1 = 0, sc = 0;
        for (int nei = 0; nei < size; nei++) {
            int uc = cas_neighbour(vc, nei, ksz, grid), ui = cas_neighbour(vi, nei, ksz, grid);
            if (uc >= 0 && T->touched[uc]) { n0 += (cov[uc] == 0); sc += cov[uc]; }
            if (ui >= 0 && T->touched[ui]) n1 += (cov[ui] == 1);
        }
        float lhs = (float)(size * (n0 - n1)), rhs = beta * (float)sc;
        if (lhs > rhs) {
            for (int nei = 0; nei < size; nei++) {
                int ui = cas_neighbour(vi, nei, ksz, grid);
                if (ui >= 0 && T->touched[ui]) cov[ui]--;
            }
            for (int nei = 0; nei < size; nei++) {
                int uc = cas_neighbour(vc, nei, ksz, grid);
                if (uc >= 0 && T->touched[uc]) cov[uc]++;
            }
            T->slot2vox[s] = vc;
        }
    }
    free(cov); free(picked);
}

static int gridify_impl(const float *data, const int *actual_numpoints, int B, int N,
                        int P, int O, int ksz, int stride, int loc,
                        const float *shift, const float *vs, const int *grid,
                        uint64_t seed, float cas_beta,
                        int *nebidx, float *nebmsk, float *cent, float *centmsk,
                        int *actual_centnum);

int gridgcn_oracle_gridify(const float *data, const int *actual_numpoints, int B, int N,
                           int P, int O, int ksz, int stride, int loc,
                           const float *shift, const float *vs, const int *grid,
                           uint64_t seed,
                           int *nebidx, float *nebmsk, float *cent, float *centmsk,
                           int *actual_centnum)
{
    return gridify_impl(data, actual_numpoints, B, N, P, O, ksz, stride, loc, shift, vs, grid, seed,
                        -1.0f, nebidx, nebmsk, cent, centmsk, actual_centnum);
}

/* Gridify with CAS (cas_refine_cloud above) between the build and the query; beta >= 0 */
int gridgcn_oracle_gridify_occaware(const float *data, const int *actual_numpoints, int B, int N,
                                    int P, int O, int ksz, int stride, int loc,
                                    const float *shift, const float *vs, const int *grid,
                                    uint64_t seed, float beta,
                                    int *nebidx, float *nebmsk, float *cent, float *centmsk,
                                    int *actual_centnum)
{
    if (!(beta >= 0.0f)) return 1;
    return gridify_impl(data, actual_numpoints, B, N, P, O, ksz, stride, loc, shift, vs, grid, seed,
                        beta, nebidx, nebmsk, cent, centmsk, actual_centnum);
}

static int gridify_impl(const float *data, const int *actual_numpoints, int B, int N,
                        int P, int O, int ksz, int stride, int loc,
                        const float *shift, const float *vs, const int *grid,
                        uint64_t seed, float cas_beta,
                        int *nebidx, float *nebmsk, float *cent, float *centmsk,
                        int *actual_centnum)
{
    (void)stride; /* accepted and ignored by the kernels */
    const int G = grid[0] * grid[1] * grid[2];
    const int size = ksz * ksz * ksz;
    const int gxy = grid[0] * grid[1];
    /* clouds are independent (every kernel indexes i_batch = index / N): OpenMP across clouds only,
     * the schedule S0 inside a cloud stays strictly sequential */
#pragma omp parallel for schedule(dynamic, 1) num_threads(oracle_threads(B))
    for (int b = 0; b < B; b++) {
        int *o_idx = nebidx + (size_t)b * O * P;
        float *o_msk = nebmsk + (size_t)b * O * P;
        float *o_cent = cent + (size_t)b * O * 4;
        float *o_cmsk = centmsk + (size_t)b * O;
        init_outputs_cloud(O, P, o_idx, o_msk, o_cent, o_cmsk, &actual_centnum[b]);
        build_tables T;
        alloc_tables(&T, G, P, O);
        if (cas_beta >= 0.0f) {
            T.order_vox = (int *)malloc((size_t)G * sizeof(int));
            T.order_idx = (int *)malloc((size_t)G * sizeof(int));
        }
        int cc = build_index_cloud(data, b, N, actual_numpoints[b], P, O, loc, shift, vs, grid,
                                   seed, &T, o_cmsk);
        if (cas_beta >= 0.0f) cas_refine_cloud(&T, cc, O, ksz, grid, seed, cas_beta);
        int cn = cc > O ? O : cc;                               /* gridify.cu:222-224 */
        actual_centnum[b] = cn;
        /* gridify_kernel_query_neighs under S0, gridify.cu:218-290 */
        for (int o = 0; o < cn; o++) {
            int index = b * O + o;
            int coor = T.slot2vox[o];                           /* :231 */
            int coor2 = coor / gxy;                             /* :232-234 (exact for G<2^24) */
            int coor1 = (coor - coor2 * gxy) / grid[0];
            int coor0 = coor - coor2 * gxy - coor1 * grid[0];
            int grid_pntidx = 0, initID = 0, origin = -1;
            float total_weight = 0.0f;
            uint32_t index_P = (uint32_t)index * (uint32_t)P;   /* :238 (int, wraps) */
            int *row = o_idx + (size_t)o * P;
            float *mrow = o_msk + (size_t)o * P;
            const float *cloud = data + (size_t)b * N * DATA_NDIM;
            for (int nei = 0; nei < size; nei++) {              /* :240 */
                int d = nei / (ksz * ksz) - (ksz - 1) / 2 + coor2;
                int h = (nei % (ksz * ksz)) / ksz - (ksz - 1) / 2 + coor1;
                int w = nei % ksz - (ksz - 1) / 2 + coor0;
                if (d >= 0 && d < grid[2] && h >= 0 && h < grid[1] && w >= 0 && w < grid[0]) {
                    int nb = d * gxy + h * grid[0] + w;
                    if (nei * 2 + 1 == size) origin = nb;       /* :248 */
                    int amount = T.cnt[nb] < P ? T.cnt[nb] : P; /* :249 */
                    for (int j = 0; j < amount; j++) {
                        if (grid_pntidx++ < P) {                /* :251 */
                            int idx = T.bucket[(size_t)nb * P + j];
                            if (grid_pntidx == 1) initID = idx;
                            int eleweight = (int)cloud[(size_t)idx * DATA_NDIM + 3]; /* :255 int trunc */
                            row[grid_pntidx - 1] = idx;
                            mrow[grid_pntidx - 1] = 1.0f;
                            total_weight += (float)eleweight;   /* :258 */
                        } else {
                            /* :260 32-bit int seed arithmetic (wraps), sign-extended */
                            int32_t s32 = (int32_t)(index_P * (uint32_t)size +
                                                    (uint32_t)grid_pntidx);
                            float u = gridgcn_oracle_xorwow_uniform((uint64_t)(int64_t)s32);
                            int r = (int)(ceilf(u * (float)grid_pntidx) - 1.0f); /* :261 */
                            if (r < P) {
                                float oldweight = cloud[(size_t)row[r] * DATA_NDIM + 3]; /* :263 */
                                int idx = T.bucket[(size_t)nb * P + j];
                                int eleweight = (int)cloud[(size_t)idx * DATA_NDIM + 3];
                                row[r] = idx;
                                total_weight += ((float)eleweight - oldweight);   /* :268 */
                            }
                        }
                    }
                }
            }
            o_cent[o * 4 + 3] = total_weight;                   /* :274 */
            if (grid_pntidx < P)
                for (int j = grid_pntidx; j < P; j++) row[j] = initID; /* :275-279 */
            if (loc == 1) {                                     /* :280-289 */
                float sw = T.sums[origin * 4 + 3];
                o_cent[o * 4 + 0] = T.sums[origin * 4 + 0] / sw;
                o_cent[o * 4 + 1] = T.sums[origin * 4 + 1] / sw;
                o_cent[o * 4 + 2] = T.sums[origin * 4 + 2] / sw;
            }
        }
        free_tables(&T);
    }
    return 0;
}

/* ------------------------------ Gridify, fast_rand build --------------------- */
/* gridifyop/fast_rand/gridify.cu under S0 (threads in ascending threadindex = index*size + j).
 * build :126-200: the thread of (point index, j) has nei_idx = (threadindex + size/2) % size and
 * appends the point to the bucket of the voxel at that offset from the point's own voxel (reservoir
 * past P seeded with curand_init(threadindex), :149-153); the thread at the zero offset also adds the
 * point to its voxel's weighted sums (loc == 1, :168-174) and claims a centre slot for a voxel seen
 * for the first time while fewer than max_o are taken (:175-197; no reservoir over the centres).
 * query :232-272: the centre reads the bucket of its own voxel only. */
int gridgcn_oracle_gridify_fast_rand(const float *data, const int *actual_numpoints, int B, int N,
                                     int P, int O, int ksz, int stride, int loc,
                                     const float *shift, const float *vs, const int *grid,
                                     int *nebidx, float *nebmsk, float *cent, float *centmsk,
                                     int *actual_centnum)
{
    (void)stride;
    const int G = grid[0] * grid[1] * grid[2];
    const int size = ksz * ksz * ksz;
    const int gxy = grid[0] * grid[1];
    if ((long long)B * N * size >= (1ll << 31)) return 1;       /* int threadindex (:126) */
#pragma omp parallel for schedule(dynamic, 1) num_threads(oracle_threads(B))
    for (int b = 0; b < B; b++) {
        int *o_idx = nebidx + (size_t)b * O * P;
        float *o_msk = nebmsk + (size_t)b * O * P;
        float *o_cent = cent + (size_t)b * O * 4;
        float *o_cmsk = centmsk + (size_t)b * O;
        init_outputs_cloud(O, P, o_idx, o_msk, o_cent, o_cmsk, &actual_centnum[b]);
        int *counter = (int *)calloc((size_t)G, sizeof(int));
        int *bucket = (int *)malloc((size_t)G * P * sizeof(int));
        float *sums = (float *)calloc((size_t)G * 4, sizeof(float));
        int *vox2idx = (int *)malloc((size_t)G * sizeof(int));
        int *slotcoor = (int *)calloc((size_t)(O > 0 ? O : 1) * 3, sizeof(int));
        for (int v = 0; v < G; v++) vox2idx[v] = -1;
        int centnum = 0;
        const float *cloud = data + (size_t)b * N * DATA_NDIM;
        for (int i = 0; i < N; i++) {
            if (!(i < actual_numpoints[b])) continue;           /* :132 */
            const float *p = cloud + (size_t)i * DATA_NDIM;
            int coor[3], inside = 1;
            for (int j = 0; j < 3; j++) {                       /* :136-142 */
                /* (the test is made on the float, as in the main build: a NaN coordinate, which
                 * CUDA's float->int conversion would turn into 0, is dropped here) */
                float f = floorf((p[j] + shift[j]) / vs[j]);
                if (!(f >= 0.0f) || !(f < (float)grid[j])) { inside = 0; break; }
                coor[j] = (int)f;
            }
            if (!inside) continue;
            for (int j = 0; j < size; j++) {
                int threadindex = (b * N + i) * size + j;       /* :126 */
                int nei_idx = (threadindex + size / 2) % size;  /* :127 */
                int d = nei_idx / (ksz * ksz) - (ksz - 1) / 2 + coor[2];       /* :144-146 */
                int h = (nei_idx % (ksz * ksz)) / ksz - (ksz - 1) / 2 + coor[1];
                int w = nei_idx % ksz - (ksz - 1) / 2 + coor[0];
                if (!(d >= 0 && d < grid[2] && h >= 0 && h < grid[1] && w >= 0 && w < grid[0]))
                    continue;
                int v = d * gxy + h * grid[0] + w;
                int c = counter[v]++;                           /* :152 */
                if (c < P) {
                    bucket[(size_t)v * P + c] = i;
                } else {                                        /* :156-160 */
                    int r = reservoir_pick((uint64_t)(int64_t)threadindex, c + 1);
                    if (r < P) bucket[(size_t)v * P + r] = i;
                }
                if (size - 1 == nei_idx * 2) {                  /* :163 the point's own voxel */
                    if (loc == 1) {                             /* :164-171 */
                        float wgt = p[3];
                        sums[v * 4 + 0] += p[0] * wgt;
                        sums[v * 4 + 1] += p[1] * wgt;
                        sums[v * 4 + 2] += p[2] * wgt;
                        sums[v * 4 + 3] += wgt;
                    }
                    if (centnum < O && vox2idx[v] == -1) {      /* :172-192 */
                        vox2idx[v] = 0;
                        int tmp = centnum++;
                        slotcoor[tmp * 3 + 0] = coor[0];
                        slotcoor[tmp * 3 + 1] = coor[1];
                        slotcoor[tmp * 3 + 2] = coor[2];
                        o_cmsk[tmp] = 1.0f;
                    }
                }
            }
        }
        actual_centnum[b] = centnum;
        for (int o = 0; o < centnum; o++) {                     /* :232-272 */
            int v = slotcoor[o * 3 + 2] * gxy + slotcoor[o * 3 + 1] * grid[0] + slotcoor[o * 3 + 0];
            int countlimit = counter[v], initID = 0;
            float xsum = 0.0f, ysum = 0.0f, zsum = 0.0f, weightsum = 0.0f, countweightsum = 0.0f;
            int *row = o_idx + (size_t)o * P;
            float *mrow = o_msk + (size_t)o * P;
            for (int j = 0; j < P; j++) {
                if (j < countlimit) {
                    int idx = bucket[(size_t)v * P + j];
                    const float *q = cloud + (size_t)idx * DATA_NDIM;
                    if (j == 0) initID = idx;
                    row[j] = idx;
                    mrow[j] = 1.0f;
                    int ew = (int)q[3];                         /* :255 int in_data_eleweight */
                    if (loc == 0) {
                        xsum += q[0] * (float)ew;
                        ysum += q[1] * (float)ew;
                        zsum += q[2] * (float)ew;
                        countweightsum += (float)ew;
                    }
                    weightsum += (float)ew;
                } else {
                    row[j] = initID;
                }
            }
            if (loc == 1) {
                xsum = sums[v * 4 + 0]; ysum = sums[v * 4 + 1]; zsum = sums[v * 4 + 2];
                countweightsum = sums[v * 4 + 3];
            }
            o_cent[o * 4 + 0] = xsum / countweightsum;
            o_cent[o * 4 + 1] = ysum / countweightsum;
            o_cent[o * 4 + 2] = zsum / countweightsum;
            o_cent[o * 4 + 3] = weightsum;
        }
        free(counter); free(bucket); free(sums); free(vox2idx); free(slotcoor);
    }
    return 0;
}

/* ------------------------------ GridifyKNN ----------------------------------- */
/* same build; query = gridifyKNN_kernel_query_neighs, gridifyknn.cu:231-332 */
int gridgcn_oracle_gridify_knn(const float *data, const int *actual_numpoints, int B, int N,
                               int P, int O, int ksz, int stride, int loc,
                               const float *shift, const float *vs, const int *grid,
                               uint64_t seed,
                               int *nebidx, float *nebmsk, float *cent, float *centmsk,
                               int *actual_centnum)
{
    (void)stride;
    if (P > 128) return 1; /* best[128] */
    const int G = grid[0] * grid[1] * grid[2];
    const int gxy = grid[0] * grid[1];
    /* clouds are independent (every kernel indexes i_batch = index / N): OpenMP across clouds only,
     * the schedule S0 inside a cloud stays strictly sequential */
#pragma omp parallel for schedule(dynamic, 1) num_threads(oracle_threads(B))
    for (int b = 0; b < B; b++) {
        int *o_idx = nebidx + (size_t)b * O * P;
        float *o_msk = nebmsk + (size_t)b * O * P;
        float *o_cent = cent + (size_t)b * O * 4;
        float *o_cmsk = centmsk + (size_t)b * O;
        init_outputs_cloud(O, P, o_idx, o_msk, o_cent, o_cmsk, &actual_centnum[b]);
        build_tables T;
        alloc_tables(&T, G, P, O);
        int cc = build_index_cloud(data, b, N, actual_numpoints[b], P, O, loc, shift, vs, grid,
                                   seed, &T, o_cmsk);
        int cn = cc > O ? O : cc;
        actual_centnum[b] = cn;
        const float *cloud = data + (size_t)b * N * DATA_NDIM;
        for (int o = 0; o < cn; o++) {
            int coor = T.slot2vox[o];
            int coor2 = coor / gxy;
            int coor1 = (coor - coor2 * gxy) / grid[0];
            int coor0 = coor - coor2 * gxy - coor1 * grid[0];
            int origin = -1;
            /* :253-255 -- (int + 0.5) is double arithmetic, coord_shift NOT subtracted */
            float ux = (float)((coor0 + 0.5) * (double)vs[0]);
            float uy = (float)((coor1 + 0.5) * (double)vs[1]);
            float uz = (float)((coor2 + 0.5) * (double)vs[2]);
            float best[128];
            int besti[128];
            for (int l = 0; l < P; l++) { best[l] = FLT_MAX; besti[l] = -1; }
            int need_P = P;
            for (int layer = 0; layer < (ksz + 1) / 2; layer++) {       /* :264 */
                int amount_layer = 0;
                for (int w = -layer; w < layer + 1; w++)
                    for (int h = -layer; h < layer + 1; h++)
                        for (int d = -layer; d < layer + 1; d++) {
                            int aw = abs(w), ah = abs(h), ad = abs(d);
                            int mx = aw > ah ? aw : ah; mx = mx > ad ? mx : ad;
                            if (mx != layer) continue;              /* :269 */
                            int dc = d + coor2, hc = h + coor1, wc = w + coor0;
                            if (!(dc >= 0 && dc < grid[2] && hc >= 0 && hc < grid[1] &&
                                  wc >= 0 && wc < grid[0])) continue;
                            int nb = dc * gxy + hc * grid[0] + wc;
                            if (layer == 0) origin = nb;
                            int amount = T.cnt[nb] < P ? T.cnt[nb] : P;
                            amount_layer += amount;
                            for (int g = 0; g < amount; g++) {
                                int idx = T.bucket[(size_t)nb * P + g];
                                float x = cloud[(size_t)idx * 4], y = cloud[(size_t)idx * 4 + 1],
                                      z = cloud[(size_t)idx * 4 + 2];
                                float dx = ux - x, dy = uy - y, dz = uz - z;
                                float dst = (dx * dx + dy * dy) + dz * dz;  /* :288 no FMA, l-to-r */
                                for (int l = 0; l < P; l++) {
                                    if (dst < best[l]) {
                                        for (int j = P - 1; j > l; j--) {
                                            best[j] = best[j - 1];
                                            besti[j] = besti[j - 1];
                                        }
                                        best[l] = dst;
                                        besti[l] = idx;
                                        break;
                                    }
                                }
                            }
                        }
                need_P -= amount_layer;                              /* :304 */
                if (need_P <= 0) break;
            }
            /* defined behaviour: unfilled besti[] == besti[0] (see header) */
            for (int l = 0; l < P; l++) if (besti[l] < 0) besti[l] = besti[0];
            float total_weight = 0.0f;
            int *row = o_idx + (size_t)o * P;
            float *mrow = o_msk + (size_t)o * P;
            for (int l = 0; l < P; l++) {                            /* :308-314 */
                row[l] = besti[l];
                int eleweight = (int)cloud[(size_t)besti[l] * 4 + 3];
                mrow[l] = 1.0f;
                total_weight += (float)eleweight;
            }
            o_cent[o * 4 + 3] = total_weight;
            if (need_P > 0)
                for (int j = P - need_P; j < P; j++) row[j] = besti[0];  /* :317-321 */
            if (loc == 1) {
                float sw = T.sums[origin * 4 + 3];
                o_cent[o * 4 + 0] = T.sums[origin * 4 + 0] / sw;
                o_cent[o * 4 + 1] = T.sums[origin * 4 + 1] / sw;
                o_cent[o * 4 + 2] = T.sums[origin * 4 + 2] / sw;
            }
        }
        free_tables(&T);
    }
    return 0;
}

/* ------------------------------ GridifyUp ------------------------------------ */
/* downdata[B,Nd,4], updata[B,O,4], down_np[B], up_np[B] -> nebidx[B,O,P] i32, nebidxmsk f32
 * gridify_up.cu:121-169 (build), :190-224 (query); output fill gridify_up-inl.h:111-112 */
int gridgcn_oracle_gridify_up(const float *downdata, const float *updata,
                              const int *down_np, const int *up_np, int B, int Nd,
                              int P, int O, int ksz,
                              const float *shift, const float *vs, const int *grid,
                              uint64_t seed, int *nebidx, float *nebmsk)
{
    const int G = grid[0] * grid[1] * grid[2];
    const int gxy = grid[0] * grid[1];
    const int size = ksz * ksz * ksz;
    memset(nebidx, 0, (size_t)B * O * P * sizeof(int));
    memset(nebmsk, 0, (size_t)B * O * P * sizeof(float));
#pragma omp parallel for schedule(dynamic, 1) num_threads(oracle_threads(B))
    for (int b = 0; b < B; b++) {
        /* per-cloud tables (the reference indexes one dense [B*G, P] table by i_batch) */
        int *cnt = (int *)calloc((size_t)G, sizeof(int));
        int *bucket = (int *)calloc((size_t)G * P, sizeof(int)); /* gridify_up.cu:284 */
        for (int i = 0; i < Nd; i++) {
            if (!(i < down_np[b])) continue;
            int index = b * Nd + i;
            const float *p = downdata + (size_t)index * DATA_NDIM;
            int c[3];
            if (voxel_of(p, shift, vs, grid, c) < 0) continue;   /* :132-138 (all k^3 threads return) */
            for (int nei = 0; nei < size; nei++) {
                int64_t threadindex = (int64_t)index * size + nei;
                int d = nei / (ksz * ksz) - (ksz - 1) / 2 + c[2];
                int h = (nei % (ksz * ksz)) / ksz - (ksz - 1) / 2 + c[1];
                int w = nei % ksz - (ksz - 1) / 2 + c[0];
                if (!(d >= 0 && d < grid[2] && h >= 0 && h < grid[1] && w >= 0 && w < grid[0]))
                    continue;
                int nb = d * gxy + h * grid[0] + w;
                int k = cnt[nb]++;                               /* :158 */
                if (k < P) {
                    bucket[(size_t)nb * P + k] = i;
                } else {
                    int r = reservoir_pick(seed + (uint64_t)threadindex, k + 1); /* :162-163 */
                    if (r < P) bucket[(size_t)nb * P + r] = i;
                }
            }
        }
        for (int o = 0; o < O; o++) {
            if (!(o < up_np[b])) continue;                       /* :194 */
            int index = b * O + o;
            const float *p = updata + (size_t)index * DATA_NDIM;
            int nb = voxel_of(p, shift, vs, grid, NULL);
            if (nb < 0) continue;
            int initID = 0; /* defined: reference leaves it uninitialised when countlimit==0 */
            int countlimit = cnt[nb];
            for (int j = 0; j < P; j++) {                        /* :212-222 */
                if (j < countlimit) {
                    int idx = bucket[(size_t)nb * P + j];
                    if (j == 0) initID = idx;
                    nebidx[(size_t)index * P + j] = idx;
                    nebmsk[(size_t)index * P + j] = 1.0f;
                } else {
                    nebidx[(size_t)index * P + j] = initID;
                }
            }
        }
        free(cnt); free(bucket);
    }
    return 0;
}

/* ------------------------------ BallKNN / KNN -------------------------------- */
/* BallKNNKernel::Map, ball_k_nn-inl.h:45-93.  rows >= upnum[b] are left untouched. */
int gridgcn_oracle_ball_knn(const float *unknown, const float *known, const int *downnum,
                            const int *upnum, int B, int n, int m, int topk, float radius,
                            int *idx_out)
{
    if (topk > 6) return 1; /* best[6] */
    float r2 = radius * radius;
    /* queries are independent: what mxnet_op::Kernel<..., cpu>::Launch does upstream */
#pragma omp parallel for schedule(static) num_threads(oracle_threads(B * n))
    for (int i = 0; i < B * n; i++) {
        int b = i / n;
        int downnum_val = downnum[b], upnum_val = upnum[b];
        if (i % n >= upnum_val) continue;
        const float *kn = known + (size_t)b * m * 3;
        const float *un = unknown + (size_t)i * 3;
        int *idx = idx_out + (size_t)i * topk;
        float ux = un[0], uy = un[1], uz = un[2];
        float best[6]; int besti[6];
        for (int l = 0; l < topk; l++) { best[l] = FLT_MAX; besti[l] = -1; }
        for (int k = 0; k < downnum_val; ++k) {
            float x = kn[k * 3 + 0], y = kn[k * 3 + 1], z = kn[k * 3 + 2];
            float dx = ux - x, dy = uy - y, dz = uz - z;
            float d = (dx * dx + dy * dy) + dz * dz;              /* :76 no FMA, l-to-r */
            if (d > r2) continue;                                 /* :77 */
            for (int l = 0; l < topk; l++) {
                if (d < best[l]) {
                    for (int j = topk - 1; j > l; j--) { best[j] = best[j - 1]; besti[j] = besti[j - 1]; }
                    best[l] = d; besti[l] = k;
                    break;
                }
            }
        }
        for (int l = 0; l < topk; l++) idx[l] = besti[l];
    }
    return 0;
}

/* KNNKernel::Map, k_nn-inl.h:42-91 (besti initialised to -1: defined behaviour) */
int gridgcn_oracle_knn(const float *unknown, const float *known, const int *downnum,
                       const int *upnum, int B, int n, int m, int topk, int *idx_out)
{
    if (topk > 64) return 1; /* the HIP operator's limit as well */
    /* queries are independent: what mxnet_op::Kernel<..., cpu>::Launch does upstream */
#pragma omp parallel for schedule(static) num_threads(oracle_threads(B * n))
    for (int i = 0; i < B * n; i++) {
        int b = i / n;
        int downnum_val = downnum[b], upnum_val = upnum[b];
        if (i % n >= upnum_val) continue;
        const float *kn = known + (size_t)b * m * 3;
        const float *un = unknown + (size_t)i * 3;
        int *idx = idx_out + (size_t)i * topk;
        float ux = un[0], uy = un[1], uz = un[2];
        float best[64]; int besti[64];                 /* per query (k_nn-inl.h: new[] per thread) */
        for (int l = 0; l < topk; l++) { best[l] = FLT_MAX; besti[l] = -1; }
        for (int k = 0; k < downnum_val; ++k) {
            float x = kn[k * 3 + 0], y = kn[k * 3 + 1], z = kn[k * 3 + 2];
            float dx = ux - x, dy = uy - y, dz = uz - z;
            float d = (dx * dx + dy * dy) + dz * dz;
            for (int l = 0; l < topk; l++) {
                if (d < best[l]) {
                    for (int j = topk - 1; j > l; j--) { best[j] = best[j - 1]; besti[j] = besti[j - 1]; }
                    best[l] = d; besti[l] = k;
                    break;
                }
            }
        }
        for (int l = 0; l < topk; l++) idx[l] = besti[l];
    }
    return 0;
}

/* batch_take_g, utils/ops.py:78-93: flat take on (B*N, C) with index + b*N, MXNet take
 * default mode='clip' (flat index clipped to [0, B*N-1]). */
int gridgcn_oracle_batch_take(const float *data, const int *index, int B, int N, int C,
                              int M /* indices per cloud */, float *out)
{
    int64_t rows = (int64_t)B * N;
    for (int b = 0; b < B; b++)
        for (int j = 0; j < M; j++) {
            int64_t flat = (int64_t)index[(size_t)b * M + j] + (int64_t)b * N;
            if (flat < 0) flat = 0;
            if (flat > rows - 1) flat = rows - 1;
            memcpy(out + ((size_t)b * M + j) * C, data + (size_t)flat * C, sizeof(float) * (size_t)C);
        }
    return 0;
}
