// gridgcn_attbwd.hip -- backward of a (32 -> C) conv + BatchNorm + ReLU layer in ONE pass over the
// edges: dX, the BatchNorm-backward sums of the layer in front AND dW (gfx950, fp32 MFMA).
//
// This is the second attention conv of GridConv (update_att_mlp2d_scnd, gcn_module_g_att.py:152:
// C/4 -> C channels, C = 128 in every up layer): its [E, C] pre-activation Z is the widest tensor of
// the step and the separate dX / dW kernels (gridgcn_direct.hip) each read it once.  Here a wave
//   1. forms dZ for a 32-row tile in the dX layout (lane = row, 16 consecutive channels; BatchNorm /
//      ReLU backward of the dense or sparse upstream gradient, as gg_k_linear_dx_direct),
//   2. feeds it to the dX MFMAs (contraction over channels) and parks it in a per-wave LDS tile,
//   3. reads it back in the transposed role -- lane = channel, rows along k -- as the B operand of
//      dW^T[cin, C] += act(Aprev)^T dZ, whose A operand is the previous layer's raw output in the
//      C/D row order, i.e. exactly the 16 values per lane the dX epilogue needs anyway.
// Z is read once, Aprev once; algorithmic bytes per edge 4(C + 2*32) instead of 4(2C + 3*32).
#include "gridgcn_mma.h"
#include "gridgcn_once.h"
#include "gridgcn_train.h"

// LDS operand layouts (round 4): every MFMA run reads its B operands with four ds_read_b128 IN FRONT of the run.
// With [step][lane] layouts each MFMA had its own ds_read_b32 and the compiler -- out of registers at 247 --
// emitted ds_read, s_waitcnt lgkmcnt(0), two MFMAs, thirty-two times per chunk: the matrix pipe waited for an LDS
// round trip at every second instruction (gridgcn_attbwd_nz.hip: 973 -> 800 us from this change alone).
//   Wl (fp32) : Wl[(chunk * 64 + lane) * WS + s] = Wdx[(chunk * 16 + s) * 64 + lane], s < 16
//   T per wave: TRANSPOSED, Tt[(cc * 32 + channel) * TS + row]: written by the row's lane one channel at a time,
//               read by the channel's lane as the rows (r & 3) + 8 (r >> 2) + 4 h -- four runs of four rows
#define GG_AF_TS 36    // LDS stride (floats) of one channel of the transposed half tile: 32 rows + 4
#define GG_AF_WS 20    // LDS stride (floats) of one lane's 16 operand values: 16 + 4 (conflict-free ds_read_b128)
#define GG_AF_TILE (64 * GG_AF_TS)

// bf16 contraction mode (gridgcn_direct.hip: gg_set_mlp_bf16): eight fp32 MFMA steps = one
// v_mfma_f32_32x32x16_bf16, operands rounded in registers (compiler-made conversion, see there)
typedef __bf16 ggaf_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ggaf_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ggaf_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned ggaf_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned ggaf_pk(float lo, float hi)
{
    const ggaf_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, ggaf_bf16x2));
}
__device__ __forceinline__ ggm_f32x16 ggaf_mfma(const ggaf_u32x4 a, const ggaf_u32x4 b, ggm_f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ggaf_bf16x8, a),
                                                   __builtin_bit_cast(ggaf_bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ float gg_af_f4(const float4 &v, int i)
{
    return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}

// NJ = C / 32 (2 or 4).  cin == ndx in {16, 32}, previous layer's BatchNorm given, C % 64 == 0.
// (second launch bound = waves per SIMD: the 64-channel form fits three -- 167 registers, no spills,
//  1.63 -> 1.23 ms on 8.4 M edges; the 128-channel form needs 247 registers and stays at two: forced
//  to three it spills 66 and gains nothing)
template <int NJ, bool BF16>
__global__ __launch_bounds__(256, (NJ == 2 && !BF16) ? 3 : 2) void gg_k_att_bwd_fused(GGLinBwd p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int C = NJ * 32;
    const int h = lane >> 5, l31 = lane & 31;
    const int cin = p.cin;
    const bool colok = l31 < cin;                      // cin = 16: half of the dX / dW^T tile is idle
    float *Wl = lds;                                   // Wdx: [C/32 chunks][64 lanes][WS]
    float *cst = lds + (C / 32) * 64 * GG_AF_WS;       // scale, shift, mean, bz, cz  [5][C]
    float *T = cst + 5 * C + wave * GG_AF_TILE;        // this wave's transposed dZ half tile [64 channels][TS]
    {
        if (BF16) {
            // [C/16 groups of 8 steps][64 lanes] x 8 bf16 in the first half of the Wl area
            for (int e = tid; e < (C / 16) * 64; e += 256) {
                const int ln = e & 63, g8 = e >> 6;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = p.Wdx[(size_t)(g8 * 8 + j) * 64 + ln];
                const ggaf_u32x4 o = {ggaf_pk(v[0], v[1]), ggaf_pk(v[2], v[3]), ggaf_pk(v[4], v[5]),
                                      ggaf_pk(v[6], v[7])};
                ((ggaf_u32x4 *)Wl)[e] = o;
            }
        } else {
            for (int i0 = tid; i0 < C * 32; i0 += 256 * 8) {      // (eight loads in flight)
                float w8[8];
#pragma unroll
                for (int u = 0; u < 8; u++) w8[u] = p.Wdx[i0 + 256 * u < C * 32 ? i0 + 256 * u : C * 32 - 1];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int i = i0 + 256 * u < C * 32 ? i0 + 256 * u : C * 32 - 1;   // (unconditional store)
                    const int st = i >> 6, ln = i & 63;
                    Wl[((st >> 4) * 64 + ln) * GG_AF_WS + (st & 15)] = w8[u];
                }
            }
        }
        for (int c = tid; c < C; c += 256) {
            const float sc = p.scale[c];
            cst[c] = sc;
            cst[C + c] = p.shift[c];
            cst[2 * C + c] = p.mean[c];
            float m1v, m2v;
            gg_bn_m12(p, c, m1v, m2v);
            cst[3 * C + c] = -(sc * p.rstd[c]) * m2v;
            cst[4 * C + c] = -(sc * m1v);
        }
    }
    __syncthreads();
    const float ps = colok ? p.pscale[l31] : 0.f, psh = colok ? p.pshift[l31] : 0.f;
    const float pm = colok ? p.pmean[l31] : 0.f, pr = colok ? p.prstd[l31] : 0.f;
    float a1 = 0.f, a2 = 0.f;
    ggm_f32x16 accw[NJ];
    ggm_zero<NJ>(accw);
    const long long ntile = (p.E + 31) >> 5;
    const bool sparse = p.amax != nullptr;
    const bool z16 = p.zfmt != 0;          // Z stored as bf16

    // row pointers of a tile: Z row, upstream-gradient row, arg-max row (dense: harmless bytes of Z,
    // never used), neighbour number of the row within its centre
    auto tileptrs = [&](long long tl, const float *&zr_, const float *&gr_, const gg_amax_t *&ar_, int &pp_) {
        long long rw = (tl << 5) + l31;
        if (rw >= p.E) rw = p.E - 1;
        zr_ = z16 ? (const float *)((const unsigned short *)p.Z + rw * C) : p.Z + rw * C;
        ar_ = (const gg_amax_t *)zr_;
        pp_ = 0;
        if (sparse) {
            const long long cen = rw / p.P;
            pp_ = (int)(rw - cen * p.P);
            gr_ = p.gval + cen * C;
            ar_ = p.amax + cen * C;
        } else {
            gr_ = p.dY + rw * p.ldy;
        }
    };
    // The 32 channels a lane consumes next (z, upstream gradient, arg-max bytes) are loaded one chunk
    // AHEAD: right after a chunk's dZ is formed its registers are free again, so the next chunk's
    // loads (the next tile's first chunk at the end of a tile) fly during the dX and dW MFMAs.
    float4 z[4], g[4];
    unsigned am[4];
    // (Z as bf16: a lane's 16 channels are two 16-byte loads; z[] then holds raw bits until zcvt())
    auto issue = [&](const float *zr_, const float *gr_, const gg_amax_t *ar_, int ci) {
        const int k0 = ci * 32 + h * 16;
        if (z16) {
            const unsigned short *zb = (const unsigned short *)zr_ + k0;
            z[0] = *(const float4 *)zb;
            z[1] = *(const float4 *)(zb + 8);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (!z16) z[q] = *(const float4 *)(zr_ + k0 + 4 * q);
            g[q] = *(const float4 *)(gr_ + k0 + 4 * q);
            am[q] = *(const unsigned *)(ar_ + k0 + 4 * q);
        }
    };
    auto zcvt = [&]() {     // bf16 pairs (low half first) -> four float4
        if (!z16) return;
        const unsigned u[8] = {__float_as_uint(z[0].x), __float_as_uint(z[0].y), __float_as_uint(z[0].z),
                               __float_as_uint(z[0].w), __float_as_uint(z[1].x), __float_as_uint(z[1].y),
                               __float_as_uint(z[1].z), __float_as_uint(z[1].w)};
#pragma unroll
        for (int q = 0; q < 4; q++)
            z[q] = make_float4(__uint_as_float(u[2 * q] << 16), __uint_as_float(u[2 * q] & 0xffff0000u),
                               __uint_as_float(u[2 * q + 1] << 16),
                               __uint_as_float(u[2 * q + 1] & 0xffff0000u));
    };
    if ((long long)blockIdx.x * 4 + wave < ntile) {
        const float *z0, *g0;
        const gg_amax_t *a0;
        int p0;
        tileptrs((long long)blockIdx.x * 4 + wave, z0, g0, a0, p0);
        issue(z0, g0, a0, 0);
    }
    for (long long tile = (long long)blockIdx.x * 4 + wave; tile < ntile;
         tile += (long long)gridDim.x * 4) {
        const long long r0 = tile << 5;
        const int nrows = (p.E - r0 < 32) ? (int)(p.E - r0) : 32;
        const float *zr, *gr;
        const gg_amax_t *ar;
        int pp;
        tileptrs(tile, zr, gr, ar, pp);
        const float *nzr = zr, *ngr = gr;              // the wave's next tile (or this one again)
        const gg_amax_t *nar = ar;
        {
            const long long tn = tile + (long long)gridDim.x * 4;
            int pn;
            if (tn < ntile) tileptrs(tn, nzr, ngr, nar, pn);
        }
        // (The arg-max bytes are loaded unconditionally and applied where the gradient is used: with
        //  the load inside `if (sparse)` the compiler closed every quad with s_waitcnt vmcnt(0).)
        // the previous layer's raw outputs in the C/D row order (rows (r&3) + 8(r>>2) + 4h, column
        // l31): A operand of the dW product and input of the epilogue's BatchNorm-backward sums
        const long long base = (r0 + 4 * h) * cin + l31;
        // zpv = NaN where the lane has no element (idle column of a 16-wide layer, row past E): the
        // activation av = max(NaN * ps + psh, 0) is then 0, so such a row adds nothing to dW whatever
        // its dZ -- no per-element row select on dZ
        float zpv[16];
        // (a second, predicate-free form of this block and of the epilogue for full tiles spilled 24-59
        //  registers and ran 10 % slower than this one)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int rr = (r & 3) + 8 * (r >> 2);
            const bool ok = colok && (nrows == 32 || rr + 4 * h < nrows);
            const float v = *(ok ? p.Aprev + base + rr * cin : p.Aprev);   // (no branch around the load)
            zpv[r] = ok ? v : __builtin_nanf("");
        }
        // (recomputed where it is used -- two FMA-and-max per pair of rows: sixteen more registers for
        //  a stored copy spilled at both widths)
        auto act2 = [&](int r) -> gg_f32x2 {
            const gg_f32x2 y = __builtin_elementwise_fma((gg_f32x2){zpv[r], zpv[r + 1]}, (gg_f32x2){ps, ps},
                                                         (gg_f32x2){psh, psh});
            return (gg_f32x2){fmaxf(y.x, 0.f), fmaxf(y.y, 0.f)};
        };
        ggm_f32x16 accx;
#pragma unroll
        for (int r = 0; r < 16; r++) accx[r] = 0.f;
        ggaf_u32x4 av8[2];
        if constexpr (BF16) {
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                float av_[8];
#pragma unroll
                for (int j = 0; j < 8; j += 2) { const gg_f32x2 t = act2(hf * 8 + j); av_[j] = t.x; av_[j + 1] = t.y; }
                av8[hf] = ggaf_u32x4{ggaf_pk(av_[0], av_[1]), ggaf_pk(av_[2], av_[3]), ggaf_pk(av_[4], av_[5]),
                                     ggaf_pk(av_[6], av_[7])};
            }
        }
        // dW^T tile j += act(Aprev)^T * dZ(columns cc*32.. of the LDS tile).  The tile belongs to this
        // wave alone: its LDS writes only have to land before its reads; the columns read here are
        // overwritten two chunks later, behind another of these barriers.
        auto dwphase = [&](int j, int cc) {
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (BF16) {
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {
                    // rows r = 8 hf + jx: two runs of four consecutive rows
                    const gg_f32x4 t0 = gg_ld_f4(T + (cc * 32 + l31) * GG_AF_TS + 16 * hf + 4 * h);
                    const gg_f32x4 t1 = gg_ld_f4(T + (cc * 32 + l31) * GG_AF_TS + 16 * hf + 8 + 4 * h);
                    const float tv[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                    const ggaf_u32x4 b8 = {ggaf_pk(tv[0], tv[1]), ggaf_pk(tv[2], tv[3]),
                                           ggaf_pk(tv[4], tv[5]), ggaf_pk(tv[6], tv[7])};
                    accw[j] = ggaf_mfma(av8[hf], b8, accw[j]);
                }
            } else {
                gg_f32x4 t4[4];
#pragma unroll
                for (int jg = 0; jg < 4; jg++) t4[jg] = gg_ld_f4(T + (cc * 32 + l31) * GG_AF_TS + 8 * jg + 4 * h);
#pragma unroll
                for (int jg = 0; jg < 4; jg++) {
                    const float tv[4] = {t4[jg].x, t4[jg].y, t4[jg].z, t4[jg].w};
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const float avr = fmaxf(__builtin_fmaf(zpv[4 * jg + i], ps, psh), 0.f);   // 0 where zpv is NaN
                        accw[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(avr, tv[i], accw[j], 0, 0, 0);
                    }
                }
            }
        };
        int s = 0;
#pragma unroll
        for (int hc = 0; hc < NJ / 2; hc++) {
#pragma unroll
            for (int cc = 0; cc < 2; cc++) {
                const int k0 = (2 * hc + cc) * 32 + h * 16;
                float4 a[4];
                __builtin_amdgcn_sched_barrier(0);     // (and the uses of the loaded chunk below the previous phase)
                zcvt();
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const gg_f32x4 d = gg_dz4v(__builtin_bit_cast(gg_f32x4, z[q]), __builtin_bit_cast(gg_f32x4, g[q]),
                                               am[q], pp, sparse, cst, C, k0 + 4 * q);
                    a[q] = __builtin_bit_cast(float4, d);
                    float *tw = T + (cc * 32 + h * 16 + 4 * q) * GG_AF_TS + l31;
                    tw[0] = d.x; tw[GG_AF_TS] = d.y; tw[2 * GG_AF_TS] = d.z; tw[3 * GG_AF_TS] = d.w;
                    __builtin_amdgcn_sched_barrier(0);   // (one quad's constants live at a time)
                }
                if (2 * hc + cc + 1 < NJ) issue(zr, gr, ar, 2 * hc + cc + 1);
                else issue(nzr, ngr, nar, 0);
                // (keep the loads HERE: left alone, the scheduler sinks them to their first use)
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (BF16) {
#pragma unroll
                    for (int g = 0; g < 2; g++) {
                        const ggaf_u32x4 a8 = {ggaf_pk(a[2 * g].x, a[2 * g].y), ggaf_pk(a[2 * g].z, a[2 * g].w),
                                               ggaf_pk(a[2 * g + 1].x, a[2 * g + 1].y),
                                               ggaf_pk(a[2 * g + 1].z, a[2 * g + 1].w)};
                        accx = ggaf_mfma(a8, ((const ggaf_u32x4 *)Wl)[(2 * (2 * hc + cc) + g) * 64 + lane], accx);
                    }
                } else {
                gg_f32x4 w4[4];
#pragma unroll
                for (int q = 0; q < 4; q++) w4[q] = gg_ld_f4(Wl + ((2 * hc + cc) * 64 + lane) * GG_AF_WS + 4 * q);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float wv[4] = {w4[q].x, w4[q].y, w4[q].z, w4[q].w};
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        accx = __builtin_amdgcn_mfma_f32_32x32x2f32(gg_af_f4(a[q], i), wv[i], accx, 0, 0, 0);
                }
                }
                // NJ = 4: dW of THIS chunk's 32 channels right away (not of both chunks after the second):
                // every chunk then has its dX and its dW MFMAs between the issue of the next chunk's
                // loads and their use.  (At NJ = 2 the same split spilled 59 registers under the
                // three-waves bound; that form keeps the two-chunk phase.)
                if constexpr (NJ == 4) dwphase(2 * hc + cc, cc);
            }
            if constexpr (NJ != 4) { dwphase(2 * hc, 0); dwphase(2 * hc + 1, 1); }
        }
        // dX tile + BatchNorm-backward sums of the previous layer
        float s1 = 0.f, s2 = 0.f;
        const float pc = -(pm * pr);                   // zhat = zp * pr + pc
        // (plain stores: with buffer stores and scalar row offsets this kernel spilled 30-80 registers)
        {
            float *xp = p.dX + base;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rr = (r & 3) + 8 * (r >> 2);
                if (colok && (nrows == 32 || rr + 4 * h < nrows)) {
                    const float dx = accx[r];
                    xp[rr * cin] = dx;
                    const float d = __builtin_fmaf(zpv[r], ps, psh) > 0.f ? dx : 0.f;
                    s1 += d;
                    s2 = __builtin_fmaf(d, __builtin_fmaf(zpv[r], pr, pc), s2);
                }
            }
        }
        a1 += s1;
        a2 += s2;
    }
    // dW^T partials: the four waves add up in LDS (fixed order), one [tile j][reg][lane] block per
    // workgroup goes to the workspace
    {
        float *blk = cst + 5 * C;                      // NJ*1024 floats over the tile area
        __syncthreads();
        for (int w = 0; w < 4; w++) {
            if (wave == w) {
#pragma unroll
                for (int j = 0; j < NJ; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int idx = (j * 16 + r) * 64 + lane;
                        blk[idx] = (w == 0 ? 0.f : blk[idx]) + accw[j][r];
                    }
            }
            __syncthreads();
        }
        float *part = p.dWpart + (size_t)blockIdx.x * NJ * 1024;
        for (int i = tid; i < NJ * 1024; i += 256) part[i] = blk[i];
    }
    __syncthreads();
    float *red = lds;                                  // [4 waves][2][32]
    {
        const float t1 = a1 + __shfl_xor(a1, 32, 64);
        const float t2 = a2 + __shfl_xor(a2, 32, 64);
        if (lane < 32) {
            red[(wave * 2 + 0) * 32 + lane] = t1;
            red[(wave * 2 + 1) * 32 + lane] = t2;
        }
    }
    __syncthreads();
    if (tid < 64) {
        const int which = tid >> 5, col = tid & 31;
        float v = 0.f;
        for (int w = 0; w < 4; w++) v += red[(w * 2 + which) * 32 + col];
        if (col < cin) atomicAdd(&p.psums[which * cin + col], (double)v);
    }
}

// dW[ch][i] = sum over workgroups of the partial D tiles: tile j, lane l, reg r hold dW^T[i][ch] with
// ch = 32j + (l & 31), i = (r & 3) + 8(r >> 2) + 4(l >> 5).  One 1024-thread workgroup per (j, r):
// 16 groups of 64 lanes each sum a slice of the waves, LDS adds the groups (deterministic order).
__global__ __launch_bounds__(1024) void gg_k_att_dw_reduce(const float *__restrict__ part, int nwaves,
                                                           int NJ, int cin, float *__restrict__ dW,
                                                           const double *__restrict__ bsums, long long E,
                                                           float *__restrict__ fm1, float *__restrict__ fm2,
                                                           float *__restrict__ fdg, float *__restrict__ fdb)
{
    __shared__ float sh[16][64];
    if (bsums && blockIdx.x == 0)       // (GGLinBwd.bsums: the BatchNorm-backward vectors of the layer)
        for (int c = threadIdx.x; c < NJ * 32; c += blockDim.x) gg_bn_bwd_fin_write(bsums, E, NJ * 32, c, fm1, fm2, fdg, fdb);
    const int j = blockIdx.x >> 4, r = blockIdx.x & 15;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    float v = 0.f;
    int w = grp;
    for (; w + 7 * 16 < nwaves; w += 8 * 16) {          // eight loads in flight, added in the same order as one by one
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) t[u] = part[((size_t)(w + 16 * u) * NJ + j) * 1024 + r * 64 + lane];
#pragma unroll
        for (int u = 0; u < 8; u++) v += t[u];
    }
    for (; w < nwaves; w += 16) v += part[((size_t)w * NJ + j) * 1024 + r * 64 + lane];
    sh[grp][lane] = v;
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
        for (int g = 0; g < 16; g++) t += sh[g][lane];
        const int ch = 32 * j + (lane & 31), i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (i < cin) dW[ch * cin + i] = t;
    }
}

static int gg_att_fused_grid(long long E, int C = 128)
{
    const long long ntile = (E + 31) >> 5;
    long long nb = (ntile + 3) / 4;
    const int per_cu = C == 128 ? 2 : 3;   // resident workgroups per CU (registers): no late third wave of them
    if (nb > 256 * per_cu) nb = 256 * per_cu;
    return (int)(nb < 1 ? 1 : nb);
}

// shapes the fused kernel takes (mirrored by gg_linear_bwd_workspace)
bool gg_att_bwd_fused_ok(long long E, int cin, int C)
{
    return (cin == 32 || cin == 16) && (C == 64 || C == 128) && E >= 32;
}

size_t gg_att_bwd_fused_workspace(long long E, int cin, int C)
{
    if (!gg_att_bwd_fused_ok(E, cin, C)) return 0;
    return (size_t)gg_att_fused_grid(E, C) * (C / 32) * 1024 * sizeof(float);
}

template <int NJ>
static int launch_att_fused(const GGLinBwd &p, hipStream_t st)
{
    static GGDevOnce attr_done;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)gg_k_att_bwd_fused<NJ, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) return 3;
        if (hipFuncSetAttribute((const void *)gg_k_att_bwd_fused<NJ, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) return 3;
        attr_done = true;
    }
    const int C = NJ * 32;
    const size_t lds = ((size_t)(C / 32) * 64 * GG_AF_WS + 5 * C + 4 * GG_AF_TILE) * sizeof(float);
    int grid = gg_att_fused_grid(p.E, C);
    if (gg_get_mlp_bf16() && grid > 512) grid = 512;   // the bf16 form holds two workgroups per CU at either width
    if (gg_get_mlp_bf16()) gg_k_att_bwd_fused<NJ, true><<<grid, 256, lds, st>>>(p);
    else gg_k_att_bwd_fused<NJ, false><<<grid, 256, lds, st>>>(p);
    if (hipGetLastError() != hipSuccess) return 3;
    gg_k_att_dw_reduce<<<NJ * 16, 1024, 0, st>>>(p.dWpart, grid, NJ, p.cin, p.dW, p.bsums, p.E, p.fin_m1,
                                                 p.fin_m2, p.fin_dgamma, p.fin_dbeta);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

// 1 = not this kernel's shape (the caller falls back to the separate dX / dW kernels)
int gg_att_bwd_fused(const GGLinBwd &p, hipStream_t st)
{
    if (!gg_att_bwd_fused_ok(p.E, p.cin, p.C)) return 1;
    if (!p.Wdx || p.ndx != p.cin || !p.dX || !p.dW || !p.dWpart || !p.pscale || !p.psums) return 1;
    if (p.cin_w != p.cin || p.rot != 0 || p.drop_thr) return 1;
    if (!p.amax && (p.ldy & 3)) return 1;
    if (p.ldz && p.ldz != p.C) return 1;
    if (p.zfmt && (p.C & 7)) return 1;
    return p.C == 64 ? launch_att_fused<2>(p, st) : launch_att_fused<4>(p, st);
}
