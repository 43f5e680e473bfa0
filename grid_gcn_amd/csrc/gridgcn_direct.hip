// gridgcn_direct.hip -- register-direct fp32 MFMA kernels of the training path (gfx950).
//
// v_mfma_f32_32x32x2_f32 wants lane l of the A operand to hold A[row l&31][k = 2s + (l>>5)] at step
// s.  A contraction does not care in which ORDER k is consumed, so instead of transposing row tiles
// through LDS a lane reads 16 consecutive floats of ITS row with four 16-byte loads -- lanes 0..31
// the first half of a 128-byte line, lanes 32..63 the second half -- and consumes them over 16 MFMA
// steps; the weights are packed (gridgcn_pack_linear, layout "Wq") in exactly that order:
//
//     chunk c (32 k's; the last one may hold 8/16/24), nq = k's of the chunk / 8,
//     step (c, q, i), q < nq, i < 4:   k = 32c + (l>>5)*4*nq + 4q + i
//
// No LDS staging of activations, no workgroup barrier in the main loop: every wave owns whole 32-row
// tiles, keeps ~100 registers, and 16 waves per CU (4 per SIMD) overlap each other's loads, MFMAs
// and stores -- the LDS-staged kernels of gridgcn_train.hip ran one wave per SIMD with the three
// phases serialised (measured), at 25-40 TFLOP/s.
#include "gridgcn_mma.h"
#include "gridgcn_once.h"
#include "gridgcn_train.h"

template <int NT> struct GGBVec { float v[NT]; };

// ---- optional bf16 contraction (process-wide switch gridgcn_set_mlp_precision) ----------------
// The register-direct kernels keep their fp32 operands in HBM and their fp32 accumulators, BatchNorm
// statistics and epilogues; only the MFMA itself changes: eight consecutive steps of the fp32
// schedule (v_mfma_f32_32x32x2_f32: 2 k's per step, lanes 0..31 one k, lanes 32..63 the other) are
// ONE v_mfma_f32_32x32x16_bf16 whose lane holds the 8 A values and the 8 B values of those steps
// (element j of the bf16x8 operand = step 8g + j): the same (lane half, k) pairing, 1/16 of the
// matrix-pipe time.  The weights are converted while a workgroup copies them into LDS.
static int g_mlp_bf16 = 0;
void gg_set_mlp_bf16(int on) { g_mlp_bf16 = on ? 1 : 0; }
int gg_get_mlp_bf16() { return g_mlp_bf16; }
// column split of the forward / dX kernels for layers of few row tiles (GRIDGCN_OPT_COL_SPLIT)
#define GG_CS_TILES 512       // at most so many 32-row tiles (16 K rows)
static int g_opt_col_split = 1;
void gg_set_col_split(int on) { g_opt_col_split = on ? 1 : 0; }
int gg_get_col_split() { return g_opt_col_split; }

typedef __bf16 ggm_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned ggm_u32x4 __attribute__((ext_vector_type(4)));

// two floats -> one register of two bf16 (round to nearest even), through the COMPILER's conversion
// (v_cvt_pk_bf16_f32): an inline-asm cvt is opaque to hipcc's hazard recogniser, which then omits
// the wait states a VALU result needs before an MFMA may read it as an operand (seen as garbage dX)
typedef __bf16 ggm_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ggm_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned gg_pk_bf16(float lo, float hi)
{
    const ggm_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, ggm_bf16x2));
}

__device__ __forceinline__ ggm_f32x16 gg_mfma_bf16(const ggm_u32x4 a, const ggm_u32x4 b, ggm_f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ggm_bf16x8, a),
                                                   __builtin_bit_cast(ggm_bf16x8, b), c, 0, 0, 0);
}

// fp32 packed weights [step][64 lanes][NV] -> LDS bf16 [group of 8 steps][64 lanes][NV] (16 bytes
// per entry), zero beyond `nsteps`
template <int NV>
__device__ __forceinline__ void gg_stage_w_bf16(ggm_u32x4 *dst, const float *__restrict__ W,
                                                int nsteps, int tid, int nthr)
{
    const int ng = (nsteps + 7) >> 3, n = ng * 64 * NV;
    // two entries = sixteen loads in flight per thread and round (see gg_stage_copy4)
    for (int e0 = tid; e0 < n; e0 += 2 * nthr) {
        float v[2][8];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int e = e0 + u * nthr < n ? e0 + u * nthr : n - 1;
            const int t = e % NV, lane = (e / NV) & 63, g = e / (NV * 64);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int st = g * 8 + j;
                const float w = W[((size_t)(st < nsteps ? st : nsteps - 1) * 64 + lane) * NV + t];
                v[u][j] = st < nsteps ? w : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int e = e0 + u * nthr;
            ggm_u32x4 o = {gg_pk_bf16(v[u][0], v[u][1]), gg_pk_bf16(v[u][2], v[u][3]), gg_pk_bf16(v[u][4], v[u][5]),
                           gg_pk_bf16(v[u][6], v[u][7])};
            dst[e < n ? e : n - 1] = o;                                          // (unconditional: see gg_stage_copy4)
        }
    }
}

// column-split staging: NV of the nvf floats of every [step][lane] entry of a packed operand, first tile t0
// (t0 % NV == 0, t0 + NV <= nvf), eight vector loads in flight per thread -- a loop of dependent single
// loads was 30-50 us of a launch whose MFMA chain takes 7
template <int NV>
__device__ __forceinline__ void gg_stage_sub(float *Wl, const float *__restrict__ W, int nent, int nvf, int t0,
                                             int tid, int nthr)
{
    typedef float vec_t __attribute__((ext_vector_type(NV)));
    for (int e0 = tid; e0 < nent; e0 += nthr * 8) {
        vec_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int e = e0 + u * nthr;
            v[u] = *(const vec_t *)(W + (size_t)(e < nent ? e : nent - 1) * nvf + t0);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int e = e0 + u * nthr;
            *(vec_t *)(Wl + (size_t)(e < nent ? e : nent - 1) * NV) = v[u];      // (unconditional: see gg_stage_copy4)
        }
    }
}

template <int NT>
__device__ __forceinline__ void gg_ldb(const float *__restrict__ base, int idx, float (&b)[NT])
{
    if constexpr (NT == 1) {
        b[0] = base[idx];
    } else if constexpr (NT == 2) {
        const float2 t = ((const float2 *)base)[idx];
        b[0] = t.x; b[1] = t.y;
    } else {
#pragma unroll
        for (int g = 0; g < NT / 4; g++) {
            const float4 t = ((const float4 *)base)[idx * (NT / 4) + g];
            b[4 * g + 0] = t.x; b[4 * g + 1] = t.y; b[4 * g + 2] = t.z; b[4 * g + 3] = t.w;
        }
    }
}

__device__ __forceinline__ float gg_f4(const float4 &v, int i)
{
    return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}

// (one FMA + one max per channel, the FMAs two channels at a time: gridgcn_mma.h)
__device__ __forceinline__ float4 gg_bnrelu4(float4 a, const float4 sc, const float4 sh)
{
    return __builtin_bit_cast(float4, gg_bnrelu4v(__builtin_bit_cast(gg_f32x4, a), __builtin_bit_cast(gg_f32x4, sc),
                                                   __builtin_bit_cast(gg_f32x4, sh)));
}

// Z[E, cout] = act(X[E, K]) * W + b, batch statistics of Z in the epilogue.  K % 8 == 0, X row
// stride K.  NT = ldw / 32 column tiles per wave (all of them: the wave owns full rows of Z).
// CS (column split, few row tiles): the layer's ldw / 32 column tiles are dealt to gridDim.y workgroup
// columns of NT tiles each -- a row tile is ONE serial chain of NT * K / 2 MFMAs per wave, and a layer of
// 2 K - 6 K rows has too few of them to fill the chip; every column group re-reads the
// rows (nothing at these sizes) and owns its columns of Z, of the bias and of the statistics.
// DROP (one column tile, fp32): Dropout of the activated input while it is loaded (GGLinFwd.drop_*).
#ifndef GG_FWD4_THREADS
#define GG_FWD4_THREADS 512    // fp32 forward at four column tiles: 8 waves = two per SIMD (768 = three per SIMD at 168 registers: 25 spilled, 372 -> 418 us at 256 -> 128)
#endif
template <int NT, bool WLDS, bool EXACT, bool BF16 = false, bool CS = false, bool DROP = false>
// (amdgpu_waves_per_eu(2): for the 256-thread column-split form -- at one wave per SIMD the compiler keeps MFMA results in
//  AGPRs and copies every value the epilogue touches; the larger workgroups are bound to <= 256 registers anyway)
__global__ __launch_bounds__(CS ? 256 : (BF16 ? (NT == 8 ? 512 : (NT == 4 ? 768 : 1024)) : (NT == 8 ? 512 : (NT == 4 ? GG_FWD4_THREADS : (NT == 2 ? 768 : 1024))))) __attribute__((amdgpu_waves_per_eu(2))) void gg_k_linear_fwd_direct(GGLinFwd p)
{
    static_assert(!DROP || (NT == 1 && !BF16 && !CS), "Dropout prologue: one column tile, fp32");
    unsigned drop_lo = p.drop_lo, drop_hi = p.drop_hi;
    if (DROP && p.drop_dev) {        // graph replay: the dropout seed advances through a device scalar
        const unsigned long long sd = (((unsigned long long)drop_hi << 32) | drop_lo) + *p.drop_dev;
        drop_lo = (unsigned)sd;
        drop_hi = (unsigned)(sd >> 32);
    }
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int K = p.K, h = lane >> 5;
    const int col0 = CS ? (int)blockIdx.y * NT * 32 : 0;
    float *Wl = lds;
    const int ng8 = (K / 2 + 7) >> 3;               // bf16: groups of 8 steps
    float *scl = lds + (BF16 ? ng8 * 64 * NT * 4 : (WLDS ? K * 32 * NT : 0));     // [K] scale, [K] shift
    if (BF16) {
        gg_stage_w_bf16<NT>((ggm_u32x4 *)Wl, p.W, K / 2, tid, blockDim.x);
    } else if (CS) {
        // NT of the ntf tiles of every [step][lane] entry of the packed operand
        gg_stage_sub<NT>(Wl, p.W, K * 32, p.ldw >> 5, (int)blockIdx.y * NT, tid, blockDim.x);
    } else if (WLDS) {
        gg_stage_copy4((float4 *)Wl, (const float4 *)p.W, K * 8 * NT, 1, tid, blockDim.x);
    }
    if (p.scale)
        for (int i = tid; i < K; i += blockDim.x) { scl[i] = p.scale[i]; scl[K + i] = p.shift[i]; }
    __syncthreads();
    const float *Wb = WLDS ? Wl : p.W;

    float ssum[NT], ssq[NT], bias[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        ssum[t] = 0.f; ssq[t] = 0.f;
        const int col = col0 + t * 32 + (lane & 31);
        bias[t] = col < p.cout ? p.b[col] : 0.f;
    }
    const long long ntile = (p.E + 31) >> 5;
    const int nfull = K >> 5, ktail = K & 31;
    // fp32 form: the NEXT chunk (32 columns: four float4 per lane, issued together so that a row's
    // 64-byte piece is fetched once) is loaded while the current one runs its 16 * NT MFMAs -- the
    // next tile's first chunk during the last chunk and the epilogue.  xa = the chunk about to be
    // consumed.
    float4 xa[4];
#pragma unroll
    for (int q = 0; q < 4; q++) xa[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto rowptrs = [&](long long tl, const float *&r1, const float *&r2) {
        long long rw = (tl << 5) + (lane & 31);
        if (rw >= p.E) rw = p.E - 1;
        r1 = p.X + rw * p.lda;
        r2 = p.X2 ? p.X2 + rw * p.lda2 - p.K1 : r1;
    };
    if (!BF16 && nfull > 0 && (long long)blockIdx.x * nw + wave < ntile) {
        const float *r1, *r2;
        rowptrs((long long)blockIdx.x * nw + wave, r1, r2);
        const float *x0 = (0 < p.K1) ? r1 : r2;
#pragma unroll
        for (int q = 0; q < 4; q++) xa[q] = *(const float4 *)(x0 + h * 16 + 4 * q);
    }
    for (long long tile = (long long)blockIdx.x * nw + wave; tile < ntile;
         tile += (long long)gridDim.x * nw) {
        const long long r0 = (long long)__builtin_amdgcn_readfirstlane((int)tile) << 5;   // (wave uniform)
        long long row = r0 + (lane & 31);
        if (row >= p.E) row = p.E - 1;
        const float *xr = p.X + row * p.lda;          // lda = row stride of X (>= K)
        // two-source rows (p.X2): columns [0, K1) come from X, [K1, K) from X2 -- the concatenation
        // the classification attention MLP consumes is never written (K1 % 32 == 0)
        const float *xr2 = p.X2 ? p.X2 + row * p.lda2 - p.K1 : xr;
        ggm_f32x16 acc[NT];
        ggm_zero<NT>(acc);
        int s = 0;
        if constexpr (BF16) {
        for (int c = 0; c < nfull; c++) {
            const int k0 = c * 32 + h * 16;
            const float *xc = (c * 32 < p.K1) ? xr : xr2;
            float4 a[4];
#pragma unroll
            for (int q = 0; q < 4; q++) a[q] = *(const float4 *)(xc + k0 + 4 * q);
            if (p.scale) {
#pragma unroll
                for (int q = 0; q < 4; q++)
                    a[q] = gg_bnrelu4(a[q], *(const float4 *)(scl + k0 + 4 * q),
                                      *(const float4 *)(scl + K + k0 + 4 * q));
            }
            {
                const ggm_u32x4 *W16 = (const ggm_u32x4 *)Wl;
#pragma unroll
                for (int g = 0; g < 2; g++) {
                    const ggm_u32x4 a8 = {gg_pk_bf16(a[2 * g].x, a[2 * g].y), gg_pk_bf16(a[2 * g].z, a[2 * g].w),
                                          gg_pk_bf16(a[2 * g + 1].x, a[2 * g + 1].y),
                                          gg_pk_bf16(a[2 * g + 1].z, a[2 * g + 1].w)};
#pragma unroll
                    for (int t = 0; t < NT; t++)
                        acc[t] = gg_mfma_bf16(a8, W16[((2 * c + g) * 64 + lane) * NT + t], acc[t]);
                }
            }
        }
        } else {
            const float *n1 = xr, *n2 = xr2;             // rows of the wave's next tile (or this one again)
            {
                const long long tn = tile + (long long)gridDim.x * nw;
                if (tn < ntile) rowptrs(tn, n1, n2);
            }
            for (int c = 0; c < nfull; c++) {
                const int k0 = c * 32 + h * 16;
                const bool last = c + 1 == nfull;
                const int cn = last ? 0 : c + 1;
                const float *xn = (cn * 32 < p.K1) ? (last ? n1 : xr) : (last ? n2 : xr2);
                float4 xb[4];
#pragma unroll
                for (int q = 0; q < 4; q++) xb[q] = *(const float4 *)(xn + cn * 32 + h * 16 + 4 * q);
                float4 a[4];
#pragma unroll
                for (int q = 0; q < 4; q++) a[q] = xa[q];
                if (p.scale) {
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        a[q] = gg_bnrelu4(a[q], *(const float4 *)(scl + k0 + 4 * q),
                                          *(const float4 *)(scl + K + k0 + 4 * q));
                }
                if constexpr (DROP) {
                    // the mask of element (row, k): the hash every reader of this activation evaluates
                    // (gg_k_bn_apply when the dropped copy is written, the dX epilogue in the backward)
                    const unsigned long long e0 = (unsigned long long)row * (unsigned long long)K + (unsigned)k0;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        float *av = &a[q].x;
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            av[i] = gg_drop_keep(e0 + (unsigned)(4 * q + i), drop_lo, drop_hi, p.drop_thr)
                                        ? av[i] * p.drop_scale : 0.f;
                    }
                }
                // the weight fragment of step st + 1 is read from LDS before the MFMAs of step st (read
                // right in front of its use, every group of NT MFMAs began with an LDS round trip)
                float bb[2][NT];
                gg_ldb<NT>(Wb, s * 64 + lane, bb[0]);
#pragma unroll
                for (int st = 0; st < 16; st++) {
                    if (st + 1 < 16) gg_ldb<NT>(Wb, (s + st + 1) * 64 + lane, bb[(st + 1) & 1]);
                    // (pinned: or the scheduler folds the two buffers back into one register set --
                    //  ds_read, s_waitcnt lgkmcnt(0), NT MFMAs, sixteen times per chunk)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < NT; t++)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(gg_f4(a[st >> 2], st & 3), bb[st & 1][t], acc[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                s += 16;
#pragma unroll
                for (int q = 0; q < 4; q++) xa[q] = xb[q];
            }
        }
        if (ktail) {
            const int nq = ktail >> 3;
            const int k0 = nfull * 32 + h * 4 * nq;
            if constexpr (BF16) {
                const ggm_u32x4 *W16 = (const ggm_u32x4 *)Wl;
                float4 at[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    at[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (q < nq) {
                        at[q] = *(const float4 *)(xr2 + k0 + 4 * q);
                        if (p.scale)
                            at[q] = gg_bnrelu4(at[q], *(const float4 *)(scl + k0 + 4 * q),
                                               *(const float4 *)(scl + K + k0 + 4 * q));
                    }
                }
#pragma unroll
                for (int g = 0; g < 2; g++) {
                    if (2 * g < nq) {
                        const ggm_u32x4 a8 = {gg_pk_bf16(at[2 * g].x, at[2 * g].y), gg_pk_bf16(at[2 * g].z, at[2 * g].w),
                                              gg_pk_bf16(at[2 * g + 1].x, at[2 * g + 1].y),
                                              gg_pk_bf16(at[2 * g + 1].z, at[2 * g + 1].w)};
#pragma unroll
                        for (int t = 0; t < NT; t++)
                            acc[t] = gg_mfma_bf16(a8, W16[((2 * nfull + g) * 64 + lane) * NT + t], acc[t]);
                    }
                }
            } else {
            for (int q = 0; q < nq; q++) {
                float4 a = *(const float4 *)(xr2 + k0 + 4 * q);
                if (p.scale)
                    a = gg_bnrelu4(a, *(const float4 *)(scl + k0 + 4 * q),
                                   *(const float4 *)(scl + K + k0 + 4 * q));
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    float b[NT];
                    gg_ldb<NT>(Wb, s * 64 + lane, b);
#pragma unroll
                    for (int t = 0; t < NT; t++)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(gg_f4(a, i), b[t], acc[t], 0, 0, 0);
                    s++;
                }
            }
            }
        }
        const int nrows = (p.E - r0 < 32) ? (int)(p.E - r0) : 32;
        const int ldz = p.ldz ? p.ldz : (EXACT ? NT * 32 : p.cout);
        if (p.rowbias) {
            // bias per group of P rows (P % 32 == 0: one group per tile): the contribution of the
            // per-centre context vector, constant over a centre's neighbours
            const float *rb = p.rowbias + (r0 / p.P) * p.cout;
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const int col = col0 + t * 32 + (lane & 31);
                bias[t] = col < p.cout ? rb[col] : 0.f;
            }
        }
        float *zp = p.Z + (r0 + 4 * h) * ldz + col0 + (lane & 31);
        unsigned short *zh = (unsigned short *)p.Z + (r0 + 4 * h) * ldz + col0 + (lane & 31);   // zfmt 1
        const bool z16 = p.zfmt != 0;
        auto zst = [&](int off, float z) {
            if (z16) zh[off] = (unsigned short)(gg_pk_bf16(z, 0.f) & 0xffffu);
            else zp[off] = z;
        };
        if (NT <= 4 && nrows == 32 && !z16) {       // (8 column tiles: this form spilled 80 registers)
            // Full row block, fp32 Z: written for instruction count (every instruction costs the MFMA
            // pipe ~5 cycles).  The block's base address is wave uniform (buffer descriptor), a lane
            // adds one constant, rows step by a scalar offset, column tiles by the immediate offset;
            // bias and statistics two rows per instruction (v_pk_add_f32 / v_pk_fma_f32).
            const gg_rsrc zs = gg_make_rsrc(p.Z ? p.Z + r0 * ldz : (float *)p.sums);
            const unsigned lo = (unsigned)(4 * h * ldz + col0 + (lane & 31)) * 4u;
            const bool wr = p.Z != nullptr;            // (nullptr: statistics only)
#pragma unroll
            for (int t = 0; t < NT; t++) {
                if (EXACT || col0 + t * 32 + (lane & 31) < p.cout) {
                    gg_f32x2 sm = {0.f, 0.f}, sq = {0.f, 0.f};
                    const gg_f32x2 b2 = {bias[t], bias[t]};
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const gg_f32x2 z = (gg_f32x2){acc[t][r], acc[t][r + 1]} + b2;
                        if (wr) {
                            gg_buf_st(z.x, zs, lo + t * 128u, (unsigned)(((r & 3) + 8 * (r >> 2)) * ldz) * 4u);
                            gg_buf_st(z.y, zs, lo + t * 128u, (unsigned)((((r + 1) & 3) + 8 * ((r + 1) >> 2)) * ldz) * 4u);
                        }
                        sm += z;
                        sq = __builtin_elementwise_fma(z, z, sq);
                    }
                    ssum[t] += sm.x + sm.y;
                    ssq[t] += sq.x + sq.y;
                }
            }
        } else if (nrows == 32) {
#pragma unroll
            for (int t = 0; t < NT; t++) {
                if (EXACT || col0 + t * 32 + (lane & 31) < p.cout) {
                    float sm = 0.f, sq = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const float z = acc[t][r] + bias[t];
                        if (p.Z) zst(((r & 3) + 8 * (r >> 2)) * ldz + t * 32, z);   // (nullptr: statistics only)
                        sm += z;
                        sq += z * z;
                    }
                    ssum[t] += sm;
                    ssq[t] += sq;
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < NT; t++) {
                float sm = 0.f, sq = 0.f;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float z = acc[t][r] + bias[t];
                    if ((EXACT || col0 + t * 32 + (lane & 31) < p.cout) && ggm_row(r, lane) < nrows) {
                        if (p.Z) zst(((r & 3) + 8 * (r >> 2)) * ldz + t * 32, z);
                        sm += z;
                        sq += z * z;
                    }
                }
                ssum[t] += sm;
                ssq[t] += sq;
            }
        }
    }
    // statistics: halves -> waves (LDS) -> one fp64 atomic per column and workgroup
    if (!p.sums) return;
    __syncthreads();
    float *red = lds;                                  // [nw][2][NT*32]
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const float sm = ssum[t] + __shfl_xor(ssum[t], 32, 64);
        const float sq = ssq[t] + __shfl_xor(ssq[t], 32, 64);
        if (lane < 32) {
            red[(wave * 2 + 0) * NT * 32 + t * 32 + lane] = sm;
            red[(wave * 2 + 1) * NT * 32 + t * 32 + lane] = sq;
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * NT * 32; i += blockDim.x) {
        const int which = i / (NT * 32), col = i - which * NT * 32;
        if (col0 + col >= p.cout) continue;
        float v = 0.f;
        for (int w = 0; w < nw; w++) v += red[(w * 2 + which) * NT * 32 + col];
        atomicAdd(&p.sums[which * p.cout + col0 + col], (double)v);
    }
    // The BatchNorm finalisation of the layer by the last workgroup to arrive -- the grid is a few hundred
    // persistent workgroups, so this is one ticket per workgroup -- instead of a 256-thread launch behind
    // every one of the ~26 layers of a step.  Hand-off in the {agent atomics on both sides} form of
    // MI355X_MICROARCH "inter-workgroup visibility": the sums above ARE agent-scope atomics, every lane
    // drains its own (s_waitcnt vmcnt(0)) before the barrier in front of the relaxed ticket, the last
    // arriver reads them back with agent-scope atomic loads.  (No agent-scope fence: that is an L2
    // write-back per workgroup on this part.)
    // The finalisation's eleven parameters are read from the kernel-argument segment HERE, through a
    // pointer the compiler cannot see through: as ordinary uses of `p` they were loaded at the top of the
    // kernel and cost the main loop 26 more spilled scalar registers.
    const GGLinFwd *kp = (const GGLinFwd *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    int *const ticket = kp->fin_ticket;
    if (!ticket) return;
    int *s_last = (int *)lds;                           // (the reduction buffer is free behind the barrier;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    //  no static LDS: the dynamic size is set to the limit)
    __syncthreads();
    if (tid == 0)
        *s_last = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                  (int)(gridDim.x * gridDim.y) - 1;
    __syncthreads();
    if (!*s_last) return;
    if (tid == 0 && kp->nbt) kp->nbt[0] += 1;           // BatchNorm1d.num_batches_tracked
    const int cout = kp->cout;
    for (int c = tid; c < cout + kp->fin_tail; c += blockDim.x) {
        if (c < cout) {
            const double s1 = __hip_atomic_load(&kp->sums[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double s2 = __hip_atomic_load(&kp->sums[cout + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            gg_bn_fin_write(s1, s2, c, kp->gamma, kp->beta, kp->E, kp->eps, kp->momentum, kp->fscale, kp->fshift,
                            kp->fmean, kp->frstd, kp->run_mean, kp->run_var);
        } else {
            // columns beyond the layer in a wider table (train_ops.RawLink): the identity
            kp->fscale[c] = 1.f; kp->fshift[c] = 0.f; kp->fmean[c] = 0.f; kp->frstd[c] = 0.f;
        }
    }
}

// few row tiles: column groups of NTS tiles over gridDim.y (see the kernel's CS form)
template <int NTS>
static int launch_fwd_direct_cs(const GGLinFwd &q, hipStream_t st)
{
    static GGDevOnce attr_done;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)gg_k_linear_fwd_direct<NTS, true, false, false, true>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 3;
        attr_done = true;
    }
    const long long ntile = (q.E + 31) >> 5;
    const int ntf = q.ldw >> 5, groups = (ntf + NTS - 1) / NTS;
    size_t lds = (size_t)q.K * 32 * NTS * 4 + (size_t)2 * q.K * 4;
    const size_t rbytes = (size_t)4 * 2 * NTS * 32 * 4;
    if (lds < rbytes) lds = rbytes;
    gg_k_linear_fwd_direct<NTS, true, false, false, true><<<dim3((unsigned)((ntile + 3) / 4), groups), 256, lds, st>>>(q);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

template <int NT>
static int launch_fwd_direct(const GGLinFwd &q, hipStream_t st)
{
    static GGDevOnce attr_done;
    if (!attr_done) {
        const void *fs[4] = {(const void *)gg_k_linear_fwd_direct<NT, true, true>,
                             (const void *)gg_k_linear_fwd_direct<NT, true, false>,
                             (const void *)gg_k_linear_fwd_direct<NT, false, true>,
                             (const void *)gg_k_linear_fwd_direct<NT, false, false>};
        for (int i = 0; i < 4; i++)
            if (hipFuncSetAttribute(fs[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 3;
        attr_done = true;
    }
    // fp32 form (next chunk prefetched: 16 more registers): 2 waves per SIMD at NT >= 4, 3 at NT = 2
    const bool use16 = g_mlp_bf16 && q.scale && (size_t)((q.K / 2 + 7) / 8) * 64 * NT * 16 + (size_t)2 * q.K * 4 <= 156 * 1024;
    // few row tiles (<= 4 per CU): 4-wave workgroups, one wave per SIMD on more CUs (see
    // launch_dx_direct)
    const long long ntile = (q.E + 31) >> 5;
    const int threads = ntile <= 1024 ? 256 :
        (use16 ? (NT == 8 ? 512 : (NT == 4 ? 768 : 1024)) : (NT == 8 ? 512 : (NT == 4 ? GG_FWD4_THREADS : (NT == 2 ? 768 : 1024))));
    const int nw = threads / 64;
    const size_t wbytes = (size_t)q.K * 32 * NT * 4, sbytes = (size_t)2 * q.K * 4;
    const size_t rbytes = (size_t)nw * 2 * NT * 32 * 4;
    long long nb = (ntile + nw - 1) / nw;
    if (nb > 256) nb = 256;
    const bool exact = q.cout == NT * 32;
    const size_t w16 = (size_t)((q.K / 2 + 7) / 8) * 64 * NT * 16;
    if (q.drop_thr) {
        // Dropout prologue: the class-score conv (one column tile, fp32, K a multiple of 32, weights in LDS)
        if constexpr (NT == 1) {
            if (g_mlp_bf16 || (q.K & 31) || wbytes + sbytes > 156 * 1024 || !q.scale || q.X2 || q.zfmt) return 1;
            static GGDevOnce attr_d;
            if (!attr_d) {
                if (hipFuncSetAttribute((const void *)gg_k_linear_fwd_direct<1, true, true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                    hipFuncSetAttribute((const void *)gg_k_linear_fwd_direct<1, true, false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                    return 3;
                attr_d = true;
            }
            size_t ld = wbytes + sbytes;
            if (ld < rbytes) ld = rbytes;
            if (exact) gg_k_linear_fwd_direct<1, true, true, false, false, true><<<(int)nb, threads, ld, st>>>(q);
            else gg_k_linear_fwd_direct<1, true, false, false, false, true><<<(int)nb, threads, ld, st>>>(q);
            return hipGetLastError() == hipSuccess ? 0 : 3;
        } else {
            return 1;
        }
    }
    // bf16 only behind a BatchNorm+ReLU (q.scale): the FIRST conv of a stack sees raw coordinates /
    // geometric features (|mean| / sigma ~ 30 for the attention inputs), which 8 mantissa bits destroy
    if (g_mlp_bf16 && q.scale && w16 + sbytes <= 156 * 1024) {
        static GGDevOnce attr16;
        if (!attr16) {
            if (hipFuncSetAttribute((const void *)gg_k_linear_fwd_direct<NT, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute((const void *)gg_k_linear_fwd_direct<NT, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return 3;
            attr16 = true;
        }
        size_t l16 = w16 + sbytes;
        if (l16 < rbytes) l16 = rbytes;
        if (exact) gg_k_linear_fwd_direct<NT, true, true, true><<<(int)nb, threads, l16, st>>>(q);
        else gg_k_linear_fwd_direct<NT, true, false, true><<<(int)nb, threads, l16, st>>>(q);
        return hipGetLastError() == hipSuccess ? 0 : 3;
    }
    // few row tiles, several column tiles (fp32 Z): the column tiles go to separate workgroups
    if (NT >= 2 && ntile <= GG_CS_TILES && !q.zfmt && g_opt_col_split) {
        if (ntile <= GG_CS_TILES / 4 || NT == 2) return launch_fwd_direct_cs<1>(q, st);
        return launch_fwd_direct_cs<2>(q, st);
    }
    const bool wlds = wbytes + sbytes <= 156 * 1024;
    size_t lds = (wlds ? wbytes : 0) + sbytes;
    if (lds < rbytes) lds = rbytes;
    if (wlds && exact) gg_k_linear_fwd_direct<NT, true, true><<<(int)nb, threads, lds, st>>>(q);
    else if (wlds) gg_k_linear_fwd_direct<NT, true, false><<<(int)nb, threads, lds, st>>>(q);
    else if (exact) gg_k_linear_fwd_direct<NT, false, true><<<(int)nb, threads, lds, st>>>(q);
    else gg_k_linear_fwd_direct<NT, false, false><<<(int)nb, threads, lds, st>>>(q);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

// p.W = Wq layout, p.K = row length of X (multiple of 8), p.ldw = 32/64/128/256
int gg_linear_fwd_direct(const GGLinFwd &p, hipStream_t st)
{
    if (p.E < 1 || p.K < 8 || (p.K & 7) || p.K > 1024) return 1;
    if (p.ldw == 32) return launch_fwd_direct<1>(p, st);
    if (p.ldw == 64) return launch_fwd_direct<2>(p, st);
    if (p.ldw == 128) return launch_fwd_direct<4>(p, st);
    if (p.ldw == 256) return launch_fwd_direct<8>(p, st);
    return 1;
}

// ------------------------------------------------------------------------------------------
// dX[E, 0:ndx] = dZ[E, C] * W[C, 0:ndx] with dZ = BatchNorm/ReLU backward of (dY, Z) formed in
// registers: a lane reads 16 consecutive channels of its row from Z and from the upstream gradient
// (dense dY, or the sparse (amax, gval) of gridgcn_pairmax_bwd) -- the same k-permutation as the
// forward kernel, with the channels as contraction index.  Epilogue: store dX and accumulate the
// BatchNorm-backward sums of the PREVIOUS layer (its raw output Aprev read in the C/D layout).
//   dz = scale*dyr - scale*m1 - scale*rstd*m2*(z - mean),   dyr = dy * (z*scale + shift > 0)
#ifndef GG_DX_RING8
#define GG_DX_RING8 3      // register sets in flight at 7-8 column tiles (128 accumulator registers)
#endif
// NT = ceil(ndx/32) column tiles, NTV = its vector width in Wdx (1/2/4/8).
// CS (column split, few row tiles): as in the forward kernel -- gridDim.y column groups of NT tiles, each
// forming dZ itself; p.dx_wstride = vector width of the packed operand (1/2/4/8).
template <int NT, bool BF16 = false, bool CS = false>
__global__ __launch_bounds__(CS ? 256 : 512) __attribute__((amdgpu_waves_per_eu(2))) void gg_k_linear_dx_direct(GGLinBwd p)
{
    constexpr int NTV = NT <= 1 ? 1 : (NT <= 2 ? 2 : (NT <= 4 ? 4 : 8));
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int C = p.C, h = lane >> 5, ldx = p.cin;
    const int dx_col0 = CS ? (int)blockIdx.y * NT * 32 : p.dx_col0;
    unsigned drop_lo = p.drop_lo, drop_hi = p.drop_hi;
    if (p.drop_thr && p.drop_dev) {  // graph replay: the dropout seed advances through a device scalar
        const unsigned long long sd = (((unsigned long long)drop_hi << 32) | drop_lo) + *p.drop_dev;
        drop_lo = (unsigned)sd;
        drop_hi = (unsigned)(sd >> 32);
    }
    float *Wl = lds;                              // [C/2 steps][64][NTV]
    const int ng8 = (C / 2 + 7) >> 3;             // bf16: groups of 8 steps, 16 bytes per entry
    float *cst = lds + (BF16 ? ng8 * 64 * NTV * 4 : C * 32 * NTV);   // scale, shift, mean, bz, cz  [5][C]
    float *pcs = cst + 5 * C;          // previous layer: scale, shift, mean, rstd  [4][NT*32] (0 beyond ndx)
    {
        if (BF16) {
            gg_stage_w_bf16<NTV>((ggm_u32x4 *)Wl, p.Wdx, C / 2, tid, blockDim.x);
        } else if (CS) {
            gg_stage_sub<NTV>(Wl, p.Wdx, C * 32, p.dx_wstride, (int)blockIdx.y * NT, tid, blockDim.x);
        } else {
            // (column-half mode, NT == 4 of a layout packed for 8 tiles: every second float4)
            gg_stage_copy4((float4 *)Wl, (const float4 *)p.Wdx + (dx_col0 ? 1 : 0), C * 8 * NTV, p.dx_wstride, tid,
                           blockDim.x);
        }
        for (int c = tid; c < C; c += blockDim.x) {
            const float sc = p.scale[c];
            cst[c] = sc;
            cst[C + c] = p.shift[c];
            cst[2 * C + c] = p.mean[c];
            float m1v, m2v;
            gg_bn_m12(p, c, m1v, m2v);
            cst[3 * C + c] = -(sc * p.rstd[c]) * m2v;
            cst[4 * C + c] = -(sc * m1v);
        }
        // (kept in LDS, not in 4*NT registers per lane: the 8-tile form spilled 70 of them)
        const bool pb = p.pscale != nullptr;
        for (int c = tid; c < NT * 32; c += blockDim.x) {
            const int col = dx_col0 + c;
            const bool ok = pb && col < p.ndx;
            pcs[c] = ok ? p.pscale[col] : 0.f;
            pcs[NT * 32 + c] = ok ? p.pshift[col] : 0.f;
            pcs[2 * NT * 32 + c] = ok ? p.pmean[col] : 0.f;
            pcs[3 * NT * 32 + c] = ok ? p.prstd[col] : 0.f;
        }
    }
    __syncthreads();
    const bool prevbn = p.pscale != nullptr;
    float a1[NT], a2[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) { a1[t] = 0.f; a2[t] = 0.f; }
    const long long ntile = (p.E + 31) >> 5;
    const int nfull = C >> 5, ktail = C & 31;
    const bool sparse = p.amax != nullptr;

    // (gridgcn_mma.h: explicit FMAs, two channels per instruction; the arg-max mask of the sparse
    //  gradient is applied by the same select as the ReLU mask)
    auto dz4 = [&](const float4 z, const float4 g, int k) -> float4 {
        return __builtin_bit_cast(float4, gg_dz4v(__builtin_bit_cast(gg_f32x4, z), __builtin_bit_cast(gg_f32x4, g),
                                                   0u, 0, false, cst, C, k));
    };
    auto dz4m = [&](const float4 z, const float4 g, unsigned am, int pp_, int k) -> float4 {
        return __builtin_bit_cast(float4, gg_dz4v(__builtin_bit_cast(gg_f32x4, z), __builtin_bit_cast(gg_f32x4, g),
                                                   am, pp_, sparse, cst, C, k));
    };

    for (long long tile = (long long)blockIdx.x * nw + wave; tile < ntile;
         tile += (long long)gridDim.x * nw) {
        // (the tile number is the same in all lanes: say so, and everything derived from it --
        //  the row block's base addresses -- is scalar)
        const long long r0 = (long long)__builtin_amdgcn_readfirstlane((int)tile) << 5;
        long long row = r0 + (lane & 31);
        if (row >= p.E) row = p.E - 1;
        const float *zr = p.Z + row * (p.ldz ? p.ldz : C);
        const float *gr;
        const gg_amax_t *ar = (const gg_amax_t *)zr;   // dense: harmless bytes, never used
        int pp = 0;
        if (sparse) {
            const long long cen = row / p.P;
            pp = (int)(row - cen * p.P);
            gr = p.gval + cen * C;
            ar = p.amax + cen * C;
        } else {
            gr = p.dY + row * p.ldy;
        }
        // no branch around the arg-max load: a conditional load made the compiler wait for ALL
        // outstanding loads (s_waitcnt vmcnt(0)) after every quad of the sparse form
        auto ldg = [&](int k) -> float4 {
            float4 g = *(const float4 *)(gr + k);
            const int4 am = gg_amax4(ar + k);
            g.x = (!sparse || am.x == pp) ? g.x : 0.f; g.y = (!sparse || am.y == pp) ? g.y : 0.f;
            g.z = (!sparse || am.z == pp) ? g.z : 0.f; g.w = (!sparse || am.w == pp) ? g.w : 0.f;
            return g;
        };
        ggm_f32x16 acc[NT];
        ggm_zero<NT>(acc);
        int s = 0;
        if constexpr (BF16) {
        for (int c = 0; c < nfull; c++) {
            const int k0 = c * 32 + h * 16;
            float4 z[4], g[4];
            unsigned amq[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                z[q] = *(const float4 *)(zr + k0 + 4 * q);
                g[q] = *(const float4 *)(gr + k0 + 4 * q);
                amq[q] = *(const unsigned *)(ar + k0 + 4 * q);    // (dense: harmless bytes, never used)
            }
            {
                // dZ formed and packed group by group (its five constants per channel quad would
                // otherwise all be live at once)
                const ggm_u32x4 *W16 = (const ggm_u32x4 *)Wl;
#pragma unroll
                for (int gq = 0; gq < 2; gq++) {
                    const float4 d0 = dz4m(z[2 * gq], g[2 * gq], amq[2 * gq], pp, k0 + 8 * gq);
                    const float4 d1 = dz4m(z[2 * gq + 1], g[2 * gq + 1], amq[2 * gq + 1], pp, k0 + 8 * gq + 4);
                    const ggm_u32x4 a8 = {gg_pk_bf16(d0.x, d0.y), gg_pk_bf16(d0.z, d0.w),
                                          gg_pk_bf16(d1.x, d1.y), gg_pk_bf16(d1.z, d1.w)};
#pragma unroll
                    for (int t = 0; t < NT; t++)
                        acc[t] = gg_mfma_bf16(a8, W16[((2 * c + gq) * 64 + lane) * NTV + t], acc[t]);
                }
            }
        }
        } else {
            // Two register sets of QS quads (a quad = 4 channels per lane = 4 MFMA steps x NT tiles),
            // dense and sparse upstream gradient alike: the loads of set u + 1 are issued, THEN set u is
            // turned into dZ and consumed.  The scheduling barriers matter as much as the second set:
            // left alone, the compiler sinks the "early" loads down to their first use (or, with a
            // copy zc = zn at the end of the body, waits for them there) -- every version of this
            // loop before round 3 ran load burst -> s_waitcnt vmcnt(0) -> MFMAs, i.e. HBM time plus
            // MFMA time.  The arg-max bytes are loaded unconditionally (dense: harmless bytes of Z) and
            // applied by the same select as the ReLU mask.
            // Round 4: at 7-8 column tiles a RING of three one-quad sets (a set is consumed in 32 MFMAs = 0.85 us,
            // less than a memory latency under load: with one set ahead the pipe idled a third of the time):
            // 547 -> 525 us at cfg4's 128 -> 256 layer.  Four sets spilled 29 registers (552 us); at 3-4 tiles a
            // ring of four two-quad sets instead of two four-quad ones was slower (306 -> 337 us).
            constexpr int QS = NT == 1 ? 2 : (NT <= 4 ? 4 : (NT <= 6 ? 2 : 1));
            constexpr int NS = NT <= 6 ? 2 : GG_DX_RING8;
            struct Set { float4 z[QS], g[QS]; unsigned am[QS]; };
            auto ldset = [&](Set &S, int u) {
#pragma unroll
                for (int j = 0; j < QS; j++) {
                    const int qi = u * QS + j;
                    const int k = (qi >> 2) * 32 + h * 16 + (qi & 3) * 4;
                    S.z[j] = *(const float4 *)(zr + k);
                    S.g[j] = *(const float4 *)(gr + k);
                    S.am[j] = *(const unsigned *)(ar + k);
                }
            };
            auto mmset = [&](const Set &S, int u) {
#pragma unroll
                for (int j = 0; j < QS; j++) {
                    const int qi = u * QS + j;
                    const int k0 = (qi >> 2) * 32 + h * 16 + (qi & 3) * 4;
                    const float4 a = dz4m(S.z[j], S.g[j], S.am[j], pp, k0);
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        float b[NTV];
                        gg_ldb<NTV>(Wl, (qi * 4 + i) * 64 + lane, b);
#pragma unroll
                        for (int t = 0; t < NT; t++)
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(gg_f4(a, i), b[t], acc[t], 0, 0, 0);
                    }
                }
            };
            const int nset = nfull * 4 / QS;
            Set S[NS];
#pragma unroll
            for (int j = 0; j < NS - 1; j++)
                if (j < nset) ldset(S[j], j);
            for (int u = 0; u < nset; u += NS) {
#pragma unroll
                for (int j = 0; j < NS; j++) {
                    if (u + j < nset) {                                      // (wave uniform)
                        const int un = u + j + NS - 1;
                        ldset(S[(j + NS - 1) % NS], un < nset ? un : nset - 1);   // (past the end: a harmless re-read)
                        __builtin_amdgcn_sched_barrier(0);
                        mmset(S[j], u + j);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            s = nfull * 16;
        }
        if (ktail) {
            const int nq = ktail >> 3;
            const int k0 = nfull * 32 + h * 4 * nq;
            if constexpr (BF16) {
                const ggm_u32x4 *W16 = (const ggm_u32x4 *)Wl;
                float4 at[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    at[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (q < nq)
                        at[q] = dz4(*(const float4 *)(zr + k0 + 4 * q), ldg(k0 + 4 * q), k0 + 4 * q);
                }
#pragma unroll
                for (int gq = 0; gq < 2; gq++) {
                    if (2 * gq < nq) {
                        const ggm_u32x4 a8 = {gg_pk_bf16(at[2 * gq].x, at[2 * gq].y), gg_pk_bf16(at[2 * gq].z, at[2 * gq].w),
                                              gg_pk_bf16(at[2 * gq + 1].x, at[2 * gq + 1].y),
                                              gg_pk_bf16(at[2 * gq + 1].z, at[2 * gq + 1].w)};
#pragma unroll
                        for (int t = 0; t < NT; t++)
                            acc[t] = gg_mfma_bf16(a8, W16[((2 * nfull + gq) * 64 + lane) * NTV + t], acc[t]);
                    }
                }
            } else {
            for (int q = 0; q < nq; q++) {
                const float4 a = dz4(*(const float4 *)(zr + k0 + 4 * q), ldg(k0 + 4 * q), k0 + 4 * q);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    float b[NTV];
                    gg_ldb<NTV>(Wl, s * 64 + lane, b);
#pragma unroll
                    for (int t = 0; t < NT; t++)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(gg_f4(a, i), b[t], acc[t], 0, 0, 0);
                    s++;
                }
            }
            }
        }
        const int nrows = (p.E - r0 < 32) ? (int)(p.E - r0) : 32;
        if (nrows == 32) {
            // Full row block: straight-line stores.  The row/column-tile part of every address is wave
            // uniform (r0 comes from a readfirstlane'd tile number) and lives in scalar registers,
            // a lane adds its constant (4h rows + its column): no per-element address arithmetic,
            // no per-row branches.  (The general form below cost ~30 VALU instructions per element:
            // 7.9 VALU per MFMA at 8 column tiles, profiles/r2_pmc_bwd_gemm.txt.)
            const gg_rsrc xs = gg_make_rsrc(p.dX + (r0 * ldx + dx_col0));
            const gg_rsrc as = gg_make_rsrc(p.Aprev + (r0 * ldx + dx_col0));
            const unsigned lo = (unsigned)(4 * h * ldx + (lane & 31)) * 4u;
            // Dropout of the input activation (the class-score conv of the head: the only call with it) in
            // this form too -- the general form below took that launch 313 us for 840 MB and 5 GFLOP.  The
            // mask is the element's hash as everywhere (gg_drop_keep of row * ldx + column).
            const bool drop = p.drop_thr != 0;
            const unsigned long long idx0 = (unsigned long long)((r0 + 4 * h) * ldx + dx_col0 + (lane & 31));
            if (drop) {
                // (an opaque copy of the row stride: its 16 * NT multiples are then formed here, in the one launch
                //  that drops, instead of being hoisted out of the tile loop into scalar registers every launch
                //  keeps live -- 238 / 384 spilled SGPRs at 4 / 8 column tiles)
                int ldx_d = ldx;
                asm volatile("" : "+s"(ldx_d));
#pragma unroll
                for (int t = 0; t < NT; t++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const unsigned long long e = idx0 + (unsigned long long)(((r & 3) + 8 * (r >> 2)) * ldx_d + t * 32);
                        acc[t][r] = gg_drop_keep(e, drop_lo, drop_hi, p.drop_thr) ? acc[t][r] * p.drop_scale : 0.f;
                    }
            }
#pragma unroll
            for (int t = 0; t < NT; t++) {
                if (dx_col0 + t * 32 + (lane & 31) < p.ndx) {
                    const bool tbn = prevbn && (p.nbn == 0 || dx_col0 + t * 32 < p.nbn);
                    if (tbn) {
                        const int pc = t * 32 + (lane & 31);
                        const float ps_t = pcs[pc], psh_t = pcs[NT * 32 + pc], pm_t = pcs[2 * NT * 32 + pc],
                                    pr_t = pcs[3 * NT * 32 + pc];
                        float zpv[16];
#pragma unroll
                        for (int r = 0; r < 16; r++)
                            zpv[r] = gg_buf_ld(as, lo + t * 128u, (unsigned)(((r & 3) + 8 * (r >> 2)) * ldx) * 4u);
                        // two rows per instruction: d = relu'(zp * ps + psh) * dx,  s1 += d,
                        // s2 += d * zhat with zhat = zp * pr + pc
                        gg_f32x2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
                        const gg_f32x2 ps2 = {ps_t, ps_t}, psh2 = {psh_t, psh_t}, pr2 = {pr_t, pr_t};
                        const float pcz = -(pm_t * pr_t);
                        const gg_f32x2 pc2 = {pcz, pcz};
#pragma unroll
                        for (int r = 0; r < 16; r += 2) {
                            const gg_f32x2 dx = {acc[t][r], acc[t][r + 1]}, zp = {zpv[r], zpv[r + 1]};
                            gg_buf_st(dx.x, xs, lo + t * 128u, (unsigned)(((r & 3) + 8 * (r >> 2)) * ldx) * 4u);
                            gg_buf_st(dx.y, xs, lo + t * 128u, (unsigned)((((r + 1) & 3) + 8 * ((r + 1) >> 2)) * ldx) * 4u);
                            const gg_f32x2 y = __builtin_elementwise_fma(zp, ps2, psh2);
                            const gg_f32x2 d = {y.x > 0.f ? dx.x : 0.f, y.y > 0.f ? dx.y : 0.f};
                            s1 += d;
                            s2 = __builtin_elementwise_fma(d, __builtin_elementwise_fma(zp, pr2, pc2), s2);
                        }
                        a1[t] += s1.x + s1.y;
                        a2[t] += s2.x + s2.y;
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; r++)
                            gg_buf_st(acc[t][r], xs, lo + t * 128u, (unsigned)(((r & 3) + 8 * (r >> 2)) * ldx) * 4u);
                    }
                }
            }
            continue;
        }
        int ldx_g = ldx;                       // (the last, partial row block: opaque stride as above)
        asm volatile("" : "+s"(ldx_g));
        const long long base = (r0 + 4 * h) * ldx_g + dx_col0 + (lane & 31);
        float *xp = p.dX + base;
        const float *ap = p.Aprev + base;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            if (dx_col0 + t * 32 + (lane & 31) < p.ndx) {
                float s1 = 0.f, s2 = 0.f;
                const int pc = t * 32 + (lane & 31);
                const float ps_t = pcs[pc], psh_t = pcs[NT * 32 + pc], pm_t = pcs[2 * NT * 32 + pc],
                            pr_t = pcs[3 * NT * 32 + pc];
                // all 16 loads of the previous layer's raw output first: dX and Aprev may alias as
                // far as the compiler knows, so loads placed between the stores were serialised
                float zpv[16];
                const bool tbn = prevbn && (p.nbn == 0 || dx_col0 + t * 32 < p.nbn);
                if (tbn) {
#pragma unroll
                    for (int r = 0; r < 16; r++) {          // (rows past the end: any valid address)
                        const int rr = (r & 3) + 8 * (r >> 2);
                        const float *a = (rr + 4 * h < nrows) ? ap + rr * ldx_g + t * 32 : p.Aprev;
                        zpv[r] = *a;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int rr = (r & 3) + 8 * (r >> 2);
                    if (rr + 4 * h < nrows) {
                        const int off = rr * ldx_g + t * 32;
                        float dx = acc[t][r];
                        if (p.drop_thr)   // the input activation went through Dropout (gg_k_bn_apply)
                            dx = gg_drop_keep((unsigned long long)(base + off), drop_lo, drop_hi,
                                              p.drop_thr) ? dx * p.drop_scale : 0.f;
                        xp[off] = dx;
                        if (tbn) {
                            const float zp = zpv[r];
                            const float d = (zp * ps_t + psh_t > 0.f) ? dx : 0.f;
                            s1 += d;
                            s2 += d * ((zp - pm_t) * pr_t);
                        }
                    }
                }
                a1[t] += s1;
                a2[t] += s2;
            }
        }
    }
    if (!prevbn) return;
    __syncthreads();
    float *red = lds;                                  // [nw][2][NT*32]
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const float s1 = a1[t] + __shfl_xor(a1[t], 32, 64);
        const float s2 = a2[t] + __shfl_xor(a2[t], 32, 64);
        if (lane < 32) {
            red[(wave * 2 + 0) * NT * 32 + t * 32 + lane] = s1;
            red[(wave * 2 + 1) * NT * 32 + t * 32 + lane] = s2;
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * NT * 32; i += blockDim.x) {
        const int which = i / (NT * 32), col = i - which * NT * 32;
        if (dx_col0 + col >= p.ndx || (p.nbn && dx_col0 + col >= p.nbn)) continue;
        float v = 0.f;
        for (int w = 0; w < nw; w++) v += red[(w * 2 + which) * NT * 32 + col];
        // (nbn > 0: psums is the [2][nbn] table of the layer that produced those columns)
        atomicAdd(&p.psums[which * (p.nbn ? p.nbn : p.cin) + dx_col0 + col], (double)v);
    }
}

template <int NT>
static int launch_dx_direct(const GGLinBwd &p, hipStream_t st)
{
    constexpr int NTV = NT <= 1 ? 1 : (NT <= 2 ? 2 : (NT <= 4 ? 4 : 8));
    static GGDevOnce attr_done;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)gg_k_linear_dx_direct<NT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 3;
        if (hipFuncSetAttribute((const void *)gg_k_linear_dx_direct<NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 3;
        attr_done = true;
    }
    // (768 threads = 3 waves per SIMD at 168 registers was measured: no gain, 59 spills)
    // narrow outputs (NT <= 2) use ~150-170 registers = 3 waves per SIMD: workgroups of 4 waves so
    // that three of them fit a CU (with 8-wave workgroups only one did)
    // few row tiles (<= 4 per CU: the layers of the coarse levels, 2 K - 6 K rows): workgroups of 4
    // waves, so that every wave has a SIMD -- and its MFMA pipe -- to itself on 2x as many CUs.  A row
    // tile is a serial chain of NT * C/2 MFMAs; 8-wave groups ran two such chains per SIMD on an
    // eighth of the chip (cfg4, the 2048 centre rows of up0 at NT = 8: 58 -> 48 us; the 6144 edge rows
    // of down2 at NT = 4: 58 -> 44 us.  What is left is the chain itself: these launches have no second
    // tile to overlap their loads with)
    const long long ntile = (p.E + 31) >> 5;
    const int threads = (NT <= 2 || ntile <= 1024) ? 256 : 512, nw = threads / 64;
    const bool bf16 = g_mlp_bf16 != 0;
    // few row tiles, several column tiles: the column tiles go to separate workgroups (a row tile is one
    // serial chain of NT * C / 2 MFMAs per wave)
    if (NT >= 2 && ntile <= GG_CS_TILES && !bf16 && p.dx_wstride == 1 && p.dx_col0 == 0 && g_opt_col_split) {
        static GGDevOnce cs_attr;
        if (!cs_attr) {
            if (hipFuncSetAttribute((const void *)gg_k_linear_dx_direct<1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute((const void *)gg_k_linear_dx_direct<2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return 3;
            cs_attr = true;
        }
        const int nts = (ntile <= GG_CS_TILES / 4 || NT == 2) ? 1 : 2;
        GGLinBwd q = p;
        q.dx_wstride = NTV;
        size_t l = ((size_t)p.C * 32 * nts + 5 * (size_t)p.C) * 4 + (size_t)4 * nts * 32 * 4;
        const size_t rb = (size_t)4 * 2 * nts * 32 * 4;
        if (l < rb) l = rb;
        const dim3 grid((unsigned)((ntile + 3) / 4), (NT + nts - 1) / nts);
        if (nts == 1) gg_k_linear_dx_direct<1, false, true><<<grid, 256, l, st>>>(q);
        else gg_k_linear_dx_direct<2, false, true><<<grid, 256, l, st>>>(q);
        return hipGetLastError() == hipSuccess ? 0 : 3;
    }
    size_t lds = (bf16 ? (size_t)((p.C / 2 + 7) / 8) * 64 * NTV * 16 + 5 * (size_t)p.C * 4
                       : ((size_t)p.C * 32 * NTV + 5 * (size_t)p.C) * 4) + (size_t)4 * NT * 32 * 4;
    const size_t rbytes = (size_t)nw * 2 * NT * 32 * 4;
    if (lds < rbytes) lds = rbytes;
    if (lds > 156 * 1024) {
        // fp32 operand of 5..8 column tiles too large for LDS (C > 151): two passes of 4 tiles over
        // the same packed operand (each stages every second float4 of it)
        if (NT <= 4 || bf16 || p.dx_wstride != 1) return 1;
        GGLinBwd q = p;
        q.dx_wstride = 2;
        q.dx_col0 = 0;
        const int rc = launch_dx_direct<4>(q, st);
        if (rc) return rc;
        q.dx_col0 = 128;
        return launch_dx_direct<4>(q, st);
    }
    // workgroups that are RESIDENT per CU (registers: 167 at NT = 1 -> three 4-wave groups; 170 at
    // NT = 2 -> two, as the bf16 form at either; > 128 for the 8-wave groups of NT >= 3 -> one): a grid
    // beyond that runs a second, partly filled round (measured on the fused attention backward:
    // 1.30 -> 1.06 ms)
    int per_cu = NT == 1 ? (bf16 ? 2 : 3) : (NT == 2 ? 2 : 1);
    while (per_cu > 1 && per_cu * lds > 152 * 1024) per_cu--;
    long long nb = (ntile + nw - 1) / nw;
    if (nb > 256 * per_cu) nb = 256 * per_cu;
    if (bf16) gg_k_linear_dx_direct<NT, true><<<(int)nb, threads, lds, st>>>(p);
    else gg_k_linear_dx_direct<NT, false><<<(int)nb, threads, lds, st>>>(p);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

// requires C % 8 == 0, Wdx packed for ndx columns, psums (if any) indexed with stride cin (nbn when nbn > 0)
int gg_linear_dx_direct(const GGLinBwd &p, hipStream_t st)
{
    if (!p.Wdx || !p.dX || p.ndx < 1 || p.ndx > 256 || p.ndx > p.cin || (p.C & 7) || p.C > 256) return 1;
    if ((p.cin & 3) && false) return 1;
    switch ((p.ndx + 31) / 32) {
    case 1: return launch_dx_direct<1>(p, st);
    case 2: return launch_dx_direct<2>(p, st);
    case 3: return launch_dx_direct<3>(p, st);
    case 4: return launch_dx_direct<4>(p, st);
    case 5: return launch_dx_direct<5>(p, st);
    case 6: return launch_dx_direct<6>(p, st);
    case 7: return launch_dx_direct<7>(p, st);
    default: return launch_dx_direct<8>(p, st);
    }
}

// ------------------------------------------------------------------------------------------
// dW[C, cin] = dZ^T[C, E] * act(Aprev)[E, cin]: the contraction runs over the ROWS, two per MFMA
// step (lanes 0..31 row 2s, lanes 32..63 row 2s+1), so both operands are read row-wise and fully
// coalesced straight into registers:
//   A operand (dZ^T): lane c' holds MT channels  mg*32*MT + MT*c' + i        (one 4/8-byte load of
//                     Z and of the upstream gradient; dZ formed with per-lane constants)
//   B operand       : lane n' holds the columns of its units -- a "quad" is 128 columns read as
//                     float4 (tile j <-> column 4n'+j), a "pair" 64 columns as float2, a "single"
//                     up to 32 columns as float (tile <-> column n')
// A wave owns MT x (4NQ + 2NP + NS) <= 10 accumulator tiles for a contiguous range of rows; the MG
// m-groups of a workgroup walk the SAME rows (the B rows hit L1/L2), RS row streams fill the rest.
// No LDS, no barriers; two register sets keep the next step's loads in flight.  Partials go to the
// workspace as [wave][tile][reg][lane]; gg_k_dw_reduce_direct sums them into the framework layout.
#ifndef GG_DW_D16
#define GG_DW_D16 6      // register sets of the 16-tile form (one wave per SIMD: depth instead of partners)
#endif
// DROP: the B operand is the DROPPED activation of the layer in front -- Dropout applied on the fly to
// relu(bn(Aprev)) (GGLinBwd.drop_*: element idx = row * cin + column, gg_drop_keep), as the dX epilogue applies
// it to the gradient -- so that no dropped copy of that activation exists (the class-score conv of the head).
template <int MT, int NQ, int NP, int NS, bool BF16 = false, bool SP = false, bool DROP = false>
__global__ __launch_bounds__((MT * (4 * NQ + 2 * NP + NS) > 10) ? 256 : 512, 1) void gg_k_linear_dw_direct(
    GGLinBwd p, int MG, int RS, long long rows_per_wg, int *__restrict__ tick, int ntick, int lds_red)
{
    constexpr int NJ = 4 * NQ + 2 * NP + NS;
    unsigned drop_lo = p.drop_lo, drop_hi = p.drop_hi;
    if (DROP && p.drop_dev) {        // graph replay: the dropout seed advances through a device scalar
        const unsigned long long sd = (((unsigned long long)drop_hi << 32) | drop_lo) + *p.drop_dev;
        drop_lo = (unsigned)sd;
        drop_hi = (unsigned)(sd >> 32);
    }
    typedef float v2f __attribute__((ext_vector_type(2)));
    // (the wave number is the same in all lanes: everything derived from it -- the wave's row range,
    //  its m-group -- lives in scalar registers)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (blockIdx.x == 0)   // tickets of the reduce kernel behind this one (stream order)
        for (int t = threadIdx.x; t < ntick; t += blockDim.x) tick[t] = 0;
    const int cq = lane & 31, h = lane >> 5;
    const int mg = wave % MG, rs = wave / MG;
    const int C = p.C, cin = p.cin;
    const int ldz = p.ldz ? p.ldz : C;
    const bool prevbn = p.pscale != nullptr;
    constexpr bool sparse = SP;

    // row range of this wave
    long long wa = (long long)blockIdx.x * rows_per_wg;
    long long wb = wa + rows_per_wg < p.E ? wa + rows_per_wg : p.E;
    long long per = ((wb - wa + RS - 1) / RS + 1) & ~1ll;
    long long ra = wa + rs * per, rb = ra + per < wb ? ra + per : wb;
    if (ra > rb) ra = rb;

    // per-lane constants.  A channel outside the layer (chok false) gets all-zero constants: its dz
    // is then 0 without a select.
    //   dz = sc * (mask ? g : 0) + ((z - mu) * bz + cz),   mask = z * sc + sh > 0  [and arg-max == p]
    const int chA = mg * 32 * MT + MT * cq;
    float sc[MT], sh[MT], mu[MT], bz[MT], cz[MT];
#pragma unroll
    for (int i = 0; i < MT; i++) {
        const int c = chA + i;
        const bool ok = c < C;
        // (unconditional loads from a clamped index, the select applied afterwards: all of them in flight together)
        const int cc = ok ? c : 0;
        const float sv = p.scale[cc], shv = p.shift[cc], muv = p.mean[cc], rsv = p.rstd[cc];
        float m1v, m2v;
        gg_bn_m12(p, cc, m1v, m2v);
        const float s = ok ? sv : 0.f;
        sc[i] = s; sh[i] = ok ? shv : 0.f; mu[i] = ok ? muv : 0.f;
        bz[i] = ok ? -(s * rsv) * m2v : 0.f;
        cz[i] = ok ? -(s * m1v) : 0.f;
    }
    const bool chok = chA + MT - 1 < C;               // C % MT == 0: all or none of the MT channels
    const int chl = chok ? chA : 0;
    int col[NJ];
    // B operand = max(x * psc + psh, lo): the previous layer's BatchNorm + ReLU (lo = 0), or x itself
    // (psc = 1, psh = 0, lo = -inf); a column outside the layer: psc = psh = 0
    float psc[NJ], psh[NJ];
    const float lo = prevbn ? 0.f : -__builtin_inff();
    {
        int j = 0;
#pragma unroll
        for (int q = 0; q < NQ; q++)
#pragma unroll
            for (int e = 0; e < 4; e++) col[j++] = q * 128 + 4 * cq + e;
#pragma unroll
        for (int q = 0; q < NP; q++)
#pragma unroll
            for (int e = 0; e < 2; e++) col[j++] = NQ * 128 + 2 * cq + e;
        if (NS) col[j++] = NQ * 128 + NP * 64 + cq;
    }
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const bool ok = col[j] < cin;
        psc[j] = ok ? (prevbn ? p.pscale[col[j]] : 1.f) : 0.f;
        psh[j] = (ok && prevbn) ? p.pshift[col[j]] : 0.f;
    }
    const bool sok = NS ? (col[NJ - 1] < cin) : false;
    const int scol = sok ? col[NJ - 1] : 0;

    struct Regs {
        float z[MT], g[MT], x[NJ];
        int am[MT], pp;
        bool ok;
        long long row;                    // (DROP only: the row of the lane's B operand)
    };
    long long cen = 0;
    int pp = 0;
    if (sparse) {
        const long long r = ra + h;
        cen = r / p.P;
        pp = (int)(r - cen * p.P);
    }
    // Branch-free on purpose: with a branch (sparse or dense, the centre-counter loop) in the body
    // the compiler closed every step with s_waitcnt vmcnt(0) and the register sets below never
    // overlapped (dW of 256 -> 128 at 41 % of the MFMA rate).
    const int Pq = sparse ? p.P : (1 << 30);
    auto load = [&](Regs &R, long long s) {
        const long long row = ra + 2 * s + h;
        R.ok = row < rb;
        const long long rw = R.ok ? row : (p.E - 1);
        if constexpr (DROP) R.row = rw;
        const float *zr = p.Z + rw * ldz + chl;
        const long long cc = R.ok ? cen : 0;
        const float *gr = sparse ? p.gval + cc * C + chl : p.dY + rw * p.ldy + chl;
        // dense: a harmless byte of Z, never used
        const gg_amax_t *ar = sparse ? p.amax + cc * C + chl : (const gg_amax_t *)zr;
        if constexpr (MT == 2) { const unsigned short t = *(const unsigned short *)ar; R.am[0] = t & 255; R.am[1] = t >> 8; }
        else R.am[0] = ar[0];
        R.pp = pp;
        pp += 2;                                   // two rows on: at most two centres further (P >= 1)
        { const bool t = pp >= Pq; pp -= t ? Pq : 0; cen += t ? 1 : 0; }
        { const bool t = pp >= Pq; pp -= t ? Pq : 0; cen += t ? 1 : 0; }
        if constexpr (MT == 2) {
            const float2 t = *(const float2 *)zr, u = *(const float2 *)gr;
            R.z[0] = t.x; R.z[1] = t.y; R.g[0] = u.x; R.g[1] = u.y;
        } else {
            R.z[0] = zr[0]; R.g[0] = gr[0];
        }
        const float *xr = p.Aprev + rw * cin;
        int j = 0;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const float4 t = *(const float4 *)(xr + q * 128 + 4 * cq);
            R.x[j++] = t.x; R.x[j++] = t.y; R.x[j++] = t.z; R.x[j++] = t.w;
        }
#pragma unroll
        for (int q = 0; q < NP; q++) {
            const float2 t = *(const float2 *)(xr + NQ * 128 + 2 * cq);
            R.x[j++] = t.x; R.x[j++] = t.y;
        }
        if (NS) R.x[j++] = xr[scol];
    };

    ggm_f32x16 acc[MT][NJ];
#pragma unroll
    for (int i = 0; i < MT; i++) ggm_zero<NJ>(acc[i]);

    // Every instruction a wave issues takes ~5 cycles away from the MFMA pipe of its SIMD
    // (tools/micro/mfma_valu.hip: fp32 MFMA and VALU do not overlap on gfx950), so the operand
    // formulas are written for instruction count: explicit FMAs, the B operand two columns at a time
    // (v_pk_fma_f32), no selects that constants can absorb.
    auto values = [&](const Regs &R, float (&dz)[MT], float (&xa)[NJ]) {
#pragma unroll
        for (int i = 0; i < MT; i++) {
            const float t = __builtin_fmaf(R.z[i] - mu[i], bz[i], cz[i]);
            bool m = __builtin_fmaf(R.z[i], sc[i], sh[i]) > 0.f;
            if (sparse) m = m && (R.am[i] == R.pp);
            const float d = __builtin_fmaf(sc[i], m ? R.g[i] : 0.f, t);
            dz[i] = R.ok ? d : 0.f;
        }
#pragma unroll
        for (int j = 0; j + 1 < NJ; j += 2) {
            const v2f y = __builtin_elementwise_fma((v2f){R.x[j], R.x[j + 1]}, (v2f){psc[j], psc[j + 1]},
                                                    (v2f){psh[j], psh[j + 1]});
            xa[j] = fmaxf(y.x, lo);
            xa[j + 1] = fmaxf(y.y, lo);
        }
        if (NJ & 1) xa[NJ - 1] = fmaxf(__builtin_fmaf(R.x[NJ - 1], psc[NJ - 1], psh[NJ - 1]), lo);
        if constexpr (DROP) {
            const unsigned long long e0 = (unsigned long long)R.row * (unsigned long long)cin;
#pragma unroll
            for (int j = 0; j < NJ; j++)
                xa[j] = gg_drop_keep(e0 + (unsigned)col[j], drop_lo, drop_hi, p.drop_thr) ? xa[j] * p.drop_scale : 0.f;
        }
    };
    auto compute = [&](const Regs &R) {
        float dz[MT], xa[NJ];
        values(R, dz, xa);
#if defined(GG_DW_ABLATE) && (GG_DW_ABLATE & 1)      // no MFMA: operands kept alive, nothing else
#pragma unroll
        for (int i = 0; i < MT; i++) asm volatile("" ::"v"(dz[i]));
#pragma unroll
        for (int j = 0; j < NJ; j++) asm volatile("" ::"v"(xa[j]));
#else
#pragma unroll
        for (int i = 0; i < MT; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(dz[i], xa[j], acc[i][j], 0, 0, 0);
#endif
    };

    // Streams of the main rounds (steps whose rows all lie inside the wave's range): the row base of
    // each stream sits in a buffer descriptor (scalar), a lane contributes one constant byte offset,
    // the step is a scalar offset -- a load is one instruction, no address arithmetic.  The sparse
    // gradient's centre moves per lane (rows 2s and 2s+1 may belong to different centres): its offset
    // is a vector register advanced by select.  stream_init(r1): r1 = first row of the next step.
    gg_rsrc rz, rx, rg, rm;
    const unsigned vz = (unsigned)(h * ldz + chl) * 4u;
    const unsigned vxq = (unsigned)(h * cin + 4 * cq) * 4u;
    const unsigned vxp = (unsigned)(h * cin + NQ * 128 + 2 * cq) * 4u;
    const unsigned vxs = (unsigned)(h * cin + scol) * 4u;
    unsigned vg = 0, va = 0, sz = 0, sx = 0, sg = 0;   // (s*: scalar byte offsets of the next step)
    long long rowv = 0;                                // (DROP: row of the lane in the next streamed step)
    const unsigned dz_ = 2u * ldz * 4u, dx_ = 2u * cin * 4u, dg_ = sparse ? 0u : 2u * p.ldy * 4u;
    const unsigned Cb = (unsigned)C;
    auto stream_init = [&](long long r1) {
        rz = gg_make_rsrc(p.Z + r1 * ldz);
        rx = gg_make_rsrc(p.Aprev + r1 * cin);
        rg = gg_make_rsrc(sparse ? p.gval : p.dY + r1 * p.ldy);
        rm = gg_make_rsrc(sparse ? (const void *)p.amax : (const void *)p.Z);
        vg = sparse ? (unsigned)(cen * C + chl) * 4u : (unsigned)(h * p.ldy + chl) * 4u;
        va = (unsigned)(cen * C + chl);
        sz = 0; sx = 0; sg = 0;
        rowv = r1 + h;
    };
    auto stream_done = [&]() {                     // the generic steps continue from here
        if (sparse) cen = (long long)((va - (unsigned)chl) / Cb);
    };
    auto load_in = [&](Regs &R) {
        R.ok = true;
        if constexpr (DROP) { R.row = rowv; rowv += 2; }
#if defined(GG_DW_ABLATE) && (GG_DW_ABLATE & 2)      // no loads in the main rounds (operands: whatever the sets hold)
        asm volatile("" : "+v"(R.z[0]), "+v"(R.g[0]), "+v"(R.x[0]), "+v"(R.x[NJ - 1]));
        return;
#endif
        if (sparse) {
            if constexpr (MT == 2) { const unsigned t = gg_buf_ld_u16(rm, va, 0); R.am[0] = t & 255; R.am[1] = t >> 8; }
            else R.am[0] = gg_buf_ld_u8(rm, va, 0);
        }
        R.pp = pp;
        if constexpr (MT == 2) {
            const gg_f32x2 t = gg_buf_ld2(rz, vz, sz), u = gg_buf_ld2(rg, vg, sg);
            R.z[0] = t.x; R.z[1] = t.y; R.g[0] = u.x; R.g[1] = u.y;
        } else {
            R.z[0] = gg_buf_ld(rz, vz, sz); R.g[0] = gg_buf_ld(rg, vg, sg);
        }
        int j = 0;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const gg_f32x4 t = gg_buf_ld4(rx, vxq + q * 512u, sx);
            R.x[j++] = t.x; R.x[j++] = t.y; R.x[j++] = t.z; R.x[j++] = t.w;
        }
#pragma unroll
        for (int q = 0; q < NP; q++) {
            const gg_f32x2 t = gg_buf_ld2(rx, vxp, sx);
            R.x[j++] = t.x; R.x[j++] = t.y;
        }
        if (NS) R.x[j++] = gg_buf_ld(rx, vxs, sx);
        sz += dz_; sx += dx_; sg += dg_;
        if (sparse) {
            pp += 2;
            const bool t1 = pp >= Pq;
            pp -= t1 ? Pq : 0;
            const bool t2 = pp >= Pq;          // (P == 1: two centres per step)
            pp -= t2 ? Pq : 0;
            const unsigned adv = (t1 ? Cb : 0u) + (t2 ? Cb : 0u);
            va += adv;
            vg += adv * 4u;
        }
    };

    const long long nsteps = (rb - ra + 1) >> 1;
    Regs A, B;
    if constexpr (BF16) {
        // eight steps (16 rows) per v_mfma_f32_32x32x16_bf16: element j of a lane's operands =
        // step 8g + j, packed two steps per register as they are formed
        const long long nin = (rb - ra) >> 1;      // steps with both rows valid
        stream_init(ra);
        for (long long s = 0; s < nsteps; s += 8) {
            unsigned dzp[MT][4], xap[NJ][4];
            const bool inside = s + 8 <= nin;      // (wave uniform) all 16 rows valid: the cheap loads
            if (!inside && s > 0 && s - 8 + 8 <= nin) stream_done();
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                float d0[MT], x0[NJ], d1[MT], x1[NJ];
                if (inside) { load_in(A); load_in(B); }
                else {
                    load(A, s + 2 * jj);      // past the end: ok = false, clamped addresses
                    load(B, s + 2 * jj + 1);
                }
                values(A, d0, x0);
                values(B, d1, x1);
#pragma unroll
                for (int i = 0; i < MT; i++) dzp[i][jj] = gg_pk_bf16(d0[i], d1[i]);
#pragma unroll
                for (int j = 0; j < NJ; j++) xap[j][jj] = gg_pk_bf16(x0[j], x1[j]);
            }
#pragma unroll
            for (int i = 0; i < MT; i++) {
                const ggm_u32x4 a8 = {dzp[i][0], dzp[i][1], dzp[i][2], dzp[i][3]};
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    const ggm_u32x4 b8 = {xap[j][0], xap[j][1], xap[j][2], xap[j][3]};
                    acc[i][j] = gg_mfma_bf16(a8, b8, acc[i][j]);
                }
            }
        }
    } else {
    // D register sets rotate: the loads of step s + D - 1 are issued before step s is consumed.  One
    // step is only MT*NJ MFMAs (64 cycles each) and a wave has at most one partner on its SIMD, so
    // a distance of one step (the first form of this loop) covered < 0.5 us of the 1-2 us a load
    // takes under traffic.  (load() past the end: ok = false, clamped addresses; called once per
    // step in ascending order, as the sparse centre counter requires.)
    // D per instantiation, from the compiler's register report: as deep as fits the occupancy the
    // launcher counts on (gg_dw_direct_cfg: 4 / 3 / 2 / 1 waves per SIMD for <= 2 / <= 4 / <= 10 /
    // 16 tiles, i.e. 128 / 168 / 256 / 512 registers) without spilling.
    constexpr int TILES = MT * NJ;
    constexpr int D = TILES == 1 ? 8
                      : TILES == 2 ? (MT == 2 ? 4 : 6)
                      : TILES <= 4 ? (MT == 2 ? 4 : (TILES == 3 ? 4 : 6))
                      : TILES == 5 ? 5
                      : TILES <= 7 ? (MT == 2 ? 6 : (TILES == 7 ? 4 : 6))
                      : TILES == 8 ? 5
                      : TILES == 9 ? 3 : (TILES == 16 ? GG_DW_D16 : 2);
    Regs R[D];
#pragma unroll
    for (int d = 0; d < D - 1; d++) load(R[d], d);
    long long s = 0;
    {
        // Rounds whose loads (steps up to s + 2D - 2) all lie inside the wave's rows.  The row base
        // of each stream sits in a buffer descriptor (scalar), a lane contributes one constant byte
        // offset, the step is a scalar offset: a load is one instruction, no address arithmetic.
        // The sparse gradient's centre moves per lane (rows 2s and 2s+1 may belong to different
        // centres): its offset is a vector register advanced by select.
        const long long nin = (rb - ra) >> 1;      // steps with both rows valid
        stream_init(ra + 2 * (D - 1));
        // (scheduling barriers: left alone, the compiler sinks all D steps' loads to the end of the
        //  loop body and consumes them together at its top -- a load burst, a wait, then 8 D MFMAs with
        //  nothing in flight: HBM time and MFMA time simply added up.  Pinned, the loads of step
        //  s + D - 1 go out in front of step s and the wait at a step's first use is vmcnt(4 (D - 1)).)
        for (; s + 2 * D - 1 <= nin; s += D) {
#pragma unroll
            for (int d = 0; d < D; d++) {
                load_in(R[(d + D - 1) % D]);
                __builtin_amdgcn_sched_barrier(0);
                compute(R[d]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        stream_done();
    }
#pragma unroll
    for (int d = 0; d < D - 1; d++)                // the sets still hold steps s .. s + D - 2
        if (s + d < nsteps) compute(R[d]);
    for (s += D - 1; s < nsteps; s++) {            // < 2D steps at the end of the wave's rows, one by one
        load(R[0], s);
        compute(R[0]);
    }
    }

    // The RS row streams of an m-group add up inside the workgroup first (LDS, fixed order rs = 1,
    // 2, ...), so that ONE partial per (workgroup, m-group) goes to the workspace: RS times fewer
    // bytes written here and read by gg_k_dw_reduce_direct (at RS = 2..4 the partials of a step were
    // ~1 GB each way).  lds_red = 0: the tiles of MG m-groups do not fit LDS -- every wave stores.
    extern __shared__ __attribute__((aligned(16))) float dwred[];
    if (lds_red && RS > 1) {
        float *slot = dwred + (size_t)mg * (MT * NJ * 1024) + lane;
        for (int r_ = 1; r_ < RS; r_++) {
            if (rs == r_) {
#pragma unroll
                for (int i = 0; i < MT; i++)
#pragma unroll
                    for (int j = 0; j < NJ; j++)
#pragma unroll
                        for (int r = 0; r < 16; r++) slot[((i * NJ + j) * 16 + r) * 64] = acc[i][j][r];
            }
            __syncthreads();
            if (rs == 0) {
#pragma unroll
                for (int i = 0; i < MT; i++)
#pragma unroll
                    for (int j = 0; j < NJ; j++)
#pragma unroll
                        for (int r = 0; r < 16; r++) acc[i][j][r] += slot[((i * NJ + j) * 16 + r) * 64];
            }
            __syncthreads();
        }
        if (rs != 0) return;
    }
    // partial: [slot][i*NJ + j][reg][lane], slot = workgroup * MG + m-group (or the global wave number)
    const long long wg = lds_red ? (long long)blockIdx.x * MG + mg
                                 : (long long)blockIdx.x * (blockDim.x >> 6) + wave;
    float *out = p.dWpart + wg * (MT * NJ * 1024) + lane;
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) out[((i * NJ + j) * 16 + r) * 64] = acc[i][j][r];
}

// dW[c][framework col] = sum over the waves of m-group mg(c) of their partial element, in a FIXED
// order (bit-reproducible), in one launch on the whole chip: grid (per/64, MG, S).  Block (x, mg, z)
// sums the z-th of S contiguous slices of the m-group's wave list for its 64 elements -- 4 thread
// slices, each walking its waves with eight loads in flight -- and, for S > 1, leaves the result in
// part2[z][mg][e]; the block that draws the last ticket of its (x, mg) column adds the S slices in
// the order z = 0 .. S-1 and writes dW.  The round-2 form gave 64 blocks of 16 slices one m-group's
// 1024 partial tiles each: 42-48 us for 16 MB.  Tickets are zeroed by the dW kernel in front.
#define GG_DWR_SL 4
__global__ __launch_bounds__(64 * GG_DWR_SL) void gg_k_dw_reduce_direct(
    const float *__restrict__ part, int nwaves, int MG, int MT, int NQ, int NP, int NS, int C, int cin,
    int cin_w, int rot, float *__restrict__ part2, int *__restrict__ tick, float *__restrict__ dW,
    const double *__restrict__ bsums, long long E, float *__restrict__ fm1, float *__restrict__ fm2,
    float *__restrict__ fdg, float *__restrict__ fdb)
{
    __shared__ float sh[64 * GG_DWR_SL];
    // (BatchNorm-backward vectors of the layer, when no launch of their own wrote them: GGLinBwd.bsums)
    if (bsums && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
        for (int c = threadIdx.x; c < C; c += blockDim.x) gg_bn_bwd_fin_write(bsums, E, C, c, fm1, fm2, fdg, fdb);
    __shared__ int s_last;
    const int NJ = 4 * NQ + 2 * NP + NS;
    const int per = MT * NJ * 1024;
    const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;          // element of the partial
    const int mg = blockIdx.y, z = blockIdx.z, S = gridDim.z;
    const int nwm = nwaves / MG;                 // waves of one m-group: w = mg + i * MG
    const int chunk = (nwm + S - 1) / S;
    const int i0 = z * chunk, i1 = i0 + chunk < nwm ? i0 + chunk : nwm;
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; u++) acc[u] = 0.f;
    if (e < per) {
        const float *src = part + (size_t)mg * per + e;
        const size_t stride = (size_t)MG * per;
        // eight loads in flight in EVERY round, the last one included (its terms beyond the slice read a valid
        // address and add zero: the same sums in the same order as a one-by-one tail, which was up to seven
        // memory round trips in a row for the short slices of the small layers)
        for (int i = i0 + sl; i < i1; i += 8 * GG_DWR_SL) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int ii = i + u * GG_DWR_SL;
                t[u] = src[(size_t)(ii < i1 ? ii : i1 - 1) * stride];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) acc[u] += (i + u * GG_DWR_SL < i1) ? t[u] : 0.f;
        }
    }
    sh[threadIdx.x] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    float s = 0.f;
    if (sl == 0) {
#pragma unroll
        for (int k = 0; k < GG_DWR_SL; k++) s += sh[k * 64 + el];
    }
    if (S > 1) {
        // hand-off without an L2 write-back: the slice sums travel as device-scope (sc1, write-through)
        // stores and are read back with device-scope loads -- "sc1 stores and loads on both sides" of
        // MI355X_MICROARCH's inter-workgroup section; a release fence per block (buffer_wbl2) made
        // this kernel 45 us with 1024 blocks in flight
        if (sl == 0 && e < per)
            __hip_atomic_store(&part2[((size_t)z * MG + mg) * per + e], s, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            s_last = __hip_atomic_fetch_add(&tick[mg * gridDim.x + blockIdx.x], 1, __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT) == S - 1;
            // (the consumer side of MI355X_MICROARCH's hand-off: one agent acquire, a barrier, then plain loads)
            if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (!s_last) return;
        if (sl == 0 && e < per) {
            // all S <= 16 slice sums requested together, added in the order z = 0 .. S-1.  (One device-scope
            // atomic load per slice, each waited for before the next was issued, was ~10 of this kernel's
            // 10 us: sixteen memory round trips in a row.)
            float v[16];
#pragma unroll
            for (int zz = 0; zz < 16; zz++)
                v[zz] = part2[((size_t)(zz < S ? zz : S - 1) * MG + mg) * per + e];
            s = 0.f;
#pragma unroll
            for (int zz = 0; zz < 16; zz++) s += zz < S ? v[zz] : 0.f;
        }
    }
    if (sl != 0 || e >= per) return;
    const int lane = e & 63, r = (e >> 6) & 15, tile = e >> 10;
    const int i = tile / NJ, j = tile - i * NJ;
    const int cq = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), nq = lane & 31;
    const int c = mg * 32 * MT + MT * cq + i;
    int n;
    if (j < 4 * NQ) n = (j >> 2) * 128 + 4 * nq + (j & 3);
    else if (j < 4 * NQ + 2 * NP) n = NQ * 128 + 2 * nq + (j - 4 * NQ);
    else n = NQ * 128 + NP * 64 + nq;
    if (c >= C || n >= cin_w || n >= cin) return;
    const int f = n < cin_w - rot ? n + rot : n - (cin_w - rot);
    dW[(size_t)c * cin_w + f] = s;
}

struct GGDwCfg { int MT, NQ, NP, NS, MG, RS, threads, nwg; long long rows_per_wg; int lds_red, slots; };

static bool gg_dw_direct_cfg(long long E, int C, int cin, GGDwCfg *c)
{
    if (cin > 320 || (cin & 3) || C > 256 || (C & 1)) return false;
    int NQ = cin / 128, rem = cin - NQ * 128;
    int NP = rem >= 64 ? 1 : 0;
    rem -= NP * 64;
    if (rem > 32) return false;   // 33..63 leftover columns: no unit reads them without overrun
    int NS = rem > 0 ? 1 : 0;
    const int NJ = 4 * NQ + 2 * NP + NS;
    int MT = (C >= 64 && NJ <= 5) ? 2 : 1;
    // 256 input columns, many rows: 2 x 8 tiles per wave, one wave per SIMD with the accumulators in
    // the AGPR half of its 512 registers -- an operand value formed by the VALU then feeds twice the
    // MFMAs (1.5 instead of 3+ VALU instructions per MFMA)
    if (NQ == 2 && NP == 0 && NS == 0 && C >= 64 && (C & 63) == 0 && E >= 65536) MT = 2;
    if ((MT * NJ > 10 && MT * NJ != 16) || NQ > 2) return false;
    const int MG = (C + 32 * MT - 1) / (32 * MT);
    if (MG > 8) return false;
    const int RS = MG >= 4 ? 1 : 4 / MG;
    c->MT = MT; c->NQ = NQ; c->NP = NP; c->NS = NS; c->MG = MG; c->RS = RS;
    c->threads = 64 * MG * RS;
    // waves per SIMD the register footprint allows (small tiles are latency bound: more waves)
    const int wps = MT * NJ <= 2 ? 4 : (MT * NJ <= 4 ? 3 : (MT * NJ <= 10 ? 2 : 1));
    int nwg = 1024 * wps / (MG * RS);
    long long maxwg = (E + 64LL * RS - 1) / (64LL * RS);
    if (nwg > maxwg) nwg = (int)maxwg;
    if (nwg < 1) nwg = 1;
    long long rp = (E + nwg - 1) / nwg;
    rp = (rp + 1) & ~1ll;
    c->rows_per_wg = rp;
    c->nwg = (int)((E + rp - 1) / rp);
    // row streams of an m-group summed in LDS before the store (one tile set per m-group must fit)
    c->lds_red = (RS > 1 && (size_t)MG * MT * NJ * 4096 <= 152 * 1024) ? 1 : 0;
    c->slots = c->lds_red ? c->nwg * MG : c->nwg * (c->threads / 64);   // partial tile sets in the workspace
    return true;
}

// slices of the reduce's wave list (grid.z), and where its scratch sits behind the partials
struct GGDwRed { int S, gx; size_t part_floats, part2_floats; };
static GGDwRed gg_dw_reduce_cfg(const GGDwCfg &c)
{
    GGDwRed r;
    const int NJ = 4 * c.NQ + 2 * c.NP + c.NS;
    const int per = c.MT * NJ * 1024;
    const int nwm = c.slots / c.MG;
    r.gx = (per + 63) / 64;
    int S = 2048 / (r.gx * c.MG);
    S = S > 16 ? 16 : S;
    while (S > 1 && nwm / S < 8) S >>= 1;          // at least 8 waves per slice
    if (nwm <= 64) S = 1;                          // two load rounds at most: no slices, no ticket, no second pass
    r.S = S < 1 ? 1 : S;
    r.part_floats = (size_t)c.slots * per;
    r.part2_floats = r.S > 1 ? (size_t)r.S * c.MG * per : 0;
    return r;
}

size_t gg_linear_dw_direct_workspace(long long E, int cin, int C)
{
    GGDwCfg c;
    if (!gg_dw_direct_cfg(E, C, cin, &c)) return 0;
    const GGDwRed r = gg_dw_reduce_cfg(c);
    return (r.part_floats + r.part2_floats) * sizeof(float) + (size_t)r.gx * c.MG * sizeof(int) + 256;
}

template <int MT, int NQ, int NP, int NS>
static int launch_dw_direct(const GGLinBwd &p, const GGDwCfg &c, hipStream_t st)
{
    const GGDwRed r = gg_dw_reduce_cfg(c);
    int *tick = (int *)(p.dWpart + r.part_floats + r.part2_floats);
    const int ntick = r.S > 1 ? r.gx * c.MG : 0;
    // (32-bit byte offsets inside a wave's row range and inside the sparse gradient)
    const long long wmax = p.ldz > p.cin ? p.ldz : p.cin;
    if (c.rows_per_wg * (wmax > p.ldy ? wmax : p.ldy) * 4 >= (1ll << 31)) return 1;
    const bool sp = p.amax != nullptr;
    if (sp && ((p.E + p.P - 1) / p.P) * (long long)p.C * 4 >= (1ll << 31)) return 1;
    const bool bf = g_mlp_bf16 && p.pscale;   // the B operand is the layer's INPUT: bf16 only behind a BatchNorm+ReLU
    const size_t ldsb = c.lds_red ? (size_t)c.MG * MT * (4 * NQ + 2 * NP + NS) * 4096 : 0;
    if (ldsb > 64 * 1024) {
        static GGDevOnce attr_done[4];
        const int vi = (bf ? 2 : 0) + (sp ? 1 : 0);
        if (!attr_done[vi]) {
            const void *f = bf ? (sp ? (const void *)gg_k_linear_dw_direct<MT, NQ, NP, NS, true, true>
                                     : (const void *)gg_k_linear_dw_direct<MT, NQ, NP, NS, true, false>)
                               : (sp ? (const void *)gg_k_linear_dw_direct<MT, NQ, NP, NS, false, true>
                                     : (const void *)gg_k_linear_dw_direct<MT, NQ, NP, NS, false, false>);
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 3;
            attr_done[vi] = true;
        }
    }
#define GG_DWL(BF, SPV)                                                                           \
    gg_k_linear_dw_direct<MT, NQ, NP, NS, BF, SPV><<<c.nwg, c.threads, ldsb, st>>>(p, c.MG, c.RS, c.rows_per_wg, tick, ntick, c.lds_red)
    if (p.drop_thr) {
        // Dropout on the B operand: the class-score conv's shape only (fp32, dense, 128 input columns)
        if constexpr (MT == 1 && NQ == 1 && NP == 0 && NS == 0) {
            if (g_mlp_bf16 || sp || !p.pscale) return 1;
            gg_k_linear_dw_direct<1, 1, 0, 0, false, false, true><<<c.nwg, c.threads, ldsb, st>>>(
                p, c.MG, c.RS, c.rows_per_wg, tick, ntick, c.lds_red);
            return hipGetLastError() == hipSuccess ? 0 : 3;
        } else {
            return 1;
        }
    }
    if (bf && sp) GG_DWL(true, true);
    else if (bf) GG_DWL(true, false);
    else if (sp) GG_DWL(false, true);
    else GG_DWL(false, false);
#undef GG_DWL
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

// dW in the framework layout [C][cin_w] (columns rotated back by p.rot).  1 = unsupported.
int gg_linear_dw_direct(const GGLinBwd &p, hipStream_t st)
{
    GGDwCfg c;
    if (!gg_dw_direct_cfg(p.E, p.C, p.cin, &c)) return 1;
    if (p.C % c.MT) return 1;
    int rc = 1;
#define GG_DWD(mt, nq, np, ns)                                                                   \
    if (c.MT == mt && c.NQ == nq && c.NP == np && c.NS == ns) rc = launch_dw_direct<mt, nq, np, ns>(p, c, st);
    GG_DWD(1, 0, 0, 1) GG_DWD(1, 0, 1, 0) GG_DWD(1, 0, 1, 1) GG_DWD(1, 1, 0, 0) GG_DWD(1, 1, 0, 1)
    GG_DWD(1, 1, 1, 0) GG_DWD(1, 1, 1, 1) GG_DWD(1, 2, 0, 0) GG_DWD(1, 2, 0, 1) GG_DWD(1, 2, 1, 0)
    GG_DWD(2, 0, 0, 1) GG_DWD(2, 0, 1, 0) GG_DWD(2, 0, 1, 1) GG_DWD(2, 1, 0, 0) GG_DWD(2, 1, 0, 1)
    GG_DWD(2, 2, 0, 0)
#undef GG_DWD
    if (rc) return rc;
    const int nwaves = c.slots;
    const GGDwRed r = gg_dw_reduce_cfg(c);
    float *part2 = p.dWpart + r.part_floats;
    int *tick = (int *)(part2 + r.part2_floats);
    gg_k_dw_reduce_direct<<<dim3(r.gx, c.MG, r.S), 64 * GG_DWR_SL, 0, st>>>(
        p.dWpart, nwaves, c.MG, c.MT, c.NQ, c.NP, c.NS, p.C, p.cin, p.cin_w, p.rot, part2, tick, p.dW,
        p.bsums, p.E, p.fin_m1, p.fin_m2, p.fin_dgamma, p.fin_dbeta);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
