// gridgcn_attbwd_nz.hip -- backward of the second attention conv of an up layer (32 -> 128 channels,
// update_att_mlp2d_scnd, gcn_module_g_att.py:152) WITHOUT its [E, 128] pre-activation Z2 (gfx950, fp32 MFMA).
//
// gg_k_att_bwd_fused (gridgcn_attbwd.hip) forms dZ2 = sc (mask ? g : 0) + (z2 - mu) bz + cz per (edge, channel)
// and therefore reads Z2 -- 1.68 GB at cfg4 up2, the widest tensor of the step, kept alive from the forward to
// here for this one reader.  But upstream of this layer is the neighbour max pool, so the first term is SPARSE
// (one arg-max edge per (centre, channel): amax, gval) and the second is AFFINE in z2 = W2 a1 + b2 with
// a1 = relu(bn1(Z1)) -- the 32-wide activation in front, which the kernel reads anyway:
//
//   dA1[e, :]  = dZ2_sparse[e, :] W2            (MFMA over the 128 channels, operand from (amax, gval) alone)
//              + a1[e, :] M + v0                (M = W2^T diag(bz) W2  [32 x 32],  v0 = W2^T (cz + bz (b2 - mu)))
//   dW2[c, :]  = sum_e dZ2_sparse[e, c] a1[e, :]                                        (MFMA over the rows)
//              + (cz_c + bz_c (b2_c - mu_c)) S1 + bz_c W2[c, :] S2      (S1 = sum_e a1[e, :], S2 = sum_e a1 a1^T)
//
// 160 MFMAs per 32-row tile instead of 128, the VALU work of the dense term gone, and 1.26 GB instead of 2.94 GB
// through the memory system: Z2 is read by nobody after the forward's max kernel and is not saved.  Same
// mathematics as the direct form up to fp32 association (tests/test_gpu_train_ops.py::test_att_bwd_noz_*: both
// paths against the float64 stock modules).
//   gg_k_att_bwd_nz        per tile: sparse dZ -> LDS tile -> dX MFMAs, dW^T MFMAs (as gg_k_att_bwd_fused), then the
//                          dense product a1 M, S2 += a1^T a1, epilogue (+ v0, BatchNorm-backward sums of layer 1, S1)
//   gg_k_att_nz_reduce     partial dW^T / S2 tiles of the workgroups -> dW (sparse part), S2
//   gg_k_att_nz_finish     dW += dense part; dgamma, dbeta, m1, m2 of this layer from its sums
#include "gridgcn_mma.h"
#include "gridgcn_once.h"
#include "gridgcn_train.h"

#define GG_NZ_TS 36    // LDS stride (floats) of one CHANNEL of a wave's transposed dZ half tile: 32 rows + 4
#define GG_NZ_WS 20    // LDS stride (floats) of one lane's 16 operand values (16 + 4: conflict-free ds_read_b128)
#define GG_NZ_C 128
#define GG_NZ_K 32

struct GGAttNz {
    const float *Z1;                     // [E][32] raw output of the layer in front
    const float *ps, *psh, *pm, *pr;     // [32] its BatchNorm: scale, shift, mean, rstd
    const float *W2, *b2;                // [128][32] (framework layout), [128]
    const float *sc, *mu, *rs;           // [128] this layer: scale = gamma * rstd, mean, rstd
    const double *bsums;                 // [2][128] BatchNorm-backward sums of this layer (gridgcn_pairmax_bwd)
    const unsigned char *amax;           // [E / P][128] arg-max neighbour
    const float *gval;                   // [E / P][128] sparse term of dZ2 there: scale * (ReLU mask ? gradient : 0)
    float *dX;                           // [E][32]
    float *part;                         // [workgroups][5][1024]
    double *psums;                       // [2][32], zero on entry: BatchNorm-backward sums of the layer in front
    double *s1;                          // [32], zero on entry: sum_e a1[e][:]
    long long E;
    int P;
};

// LDS operand layouts: every MFMA run of this kernel reads its B operands with four ds_read_b128 IN FRONT of the
// run (one ds_read_b32 + s_waitcnt in front of every second MFMA, as a [step][lane] layout gives, left the matrix
// pipe waiting for an LDS round trip 32 times per chunk):
//   Wl[(chunk * 64 + lane) * WS + s] = W2[k][col],  k = 32 chunk + 16 (lane >> 5) + s, col = lane & 31, s < 16
//   Mp[lane * WS + s]                = M[j][col],   j = 16 (lane >> 5) + s
//   T (per wave, TRANSPOSED): Tt[(cc * 32 + ch) * TS + row] -- written by the row's lane one channel at a time,
//       read by the channel's lane as the rows (r & 3) + 8 (r >> 2) + 4 h: four runs of four consecutive rows
__device__ __forceinline__ float gg_nz_w2(const float *Wl, int c, int i)
{
    return Wl[((c >> 5) * 64 + ((c >> 4) & 1) * 32 + i) * GG_NZ_WS + (c & 15)];
}

__global__ __launch_bounds__(256, 2) void gg_k_att_bwd_nz(GGAttNz p)
{
    constexpr int C = GG_NZ_C, NJ = 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    float *Wl = lds;                       // [4 chunks][64 lanes][WS]   dX operand
    float *Mp = Wl + 4 * 64 * GG_NZ_WS;    // [64 lanes][WS]             dense operand M
    float *csc = Mp + 64 * GG_NZ_WS;       // [C] scale
    float *cbz = csc + C;                  // [C] bz = -(sc rstd) m2
    float *ct = cbz + C;                   // [C] cz + bz (b2 - mu)
    float *v0 = ct + C;                    // [32]
    float *pcs = v0 + 32;                  // [2][32] previous layer's scale, shift
    float *T = pcs + 64 + wave * (64 * GG_NZ_TS);
    for (int i0 = tid; i0 < C * 32; i0 += 256 * 8) {  // (eight loads in flight: one by one they were 16 L2 round trips)
        float w8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) w8[u] = p.W2[i0 + 256 * u < C * 32 ? i0 + 256 * u : C * 32 - 1];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + 256 * u < C * 32 ? i0 + 256 * u : C * 32 - 1;   // (unconditional store: a load whose
            const int k = i >> 5, col = i & 31;       //  only use is conditional is sunk into the branch and waited for)
            Wl[((k >> 5) * 64 + ((k >> 4) & 1) * 32 + col) * GG_NZ_WS + (k & 15)] = w8[u];
        }
    }
    for (int c = tid; c < C; c += 256) {
        const float sc = p.sc[c];
        const float m1 = (float)(p.bsums[c] / (double)p.E), m2 = (float)(p.bsums[C + c] / (double)p.E);
        const float bz = -(sc * p.rs[c]) * m2, cz = -(sc * m1);
        csc[c] = sc;
        cbz[c] = bz;
        ct[c] = cz + bz * (p.b2[c] - p.mu[c]);
    }
    if (tid < 32) { pcs[tid] = p.ps[tid]; pcs[32 + tid] = p.psh[tid]; }
    __syncthreads();
    for (int i = tid; i < 1024; i += 256) {
        const int s = i >> 6, ln = i & 63, j = 16 * (ln >> 5) + s, col = ln & 31;
        float m = 0.f;
        for (int c = 0; c < C; c++) m = __builtin_fmaf(cbz[c] * gg_nz_w2(Wl, c, j), gg_nz_w2(Wl, c, col), m);
        Mp[ln * GG_NZ_WS + s] = m;
    }
    if (tid < 32) {
        float v = 0.f;
        for (int c = 0; c < C; c++) v = __builtin_fmaf(ct[c], gg_nz_w2(Wl, c, tid), v);
        v0[tid] = v;
    }
    __syncthreads();
    const float ps = pcs[l31], psh = pcs[32 + l31];
    const float pm = p.pm[l31], pr = p.pr[l31];
    const float pc = -(pm * pr);                       // zhat = zp * pr + pc
    const float v0l = v0[l31];
    float a1 = 0.f, a2 = 0.f, a3 = 0.f;
    ggm_f32x16 accw[NJ], accS;
    ggm_zero<NJ>(accw);
#pragma unroll
    for (int r = 0; r < 16; r++) accS[r] = 0.f;
    const long long ntile = (p.E + 31) >> 5;

    auto tileptrs = [&](long long tl, const float *&gr_, const unsigned char *&ar_, int &pp_) {
        long long rw = (tl << 5) + l31;
        if (rw >= p.E) rw = p.E - 1;
        const long long cen = rw / p.P;
        pp_ = (int)(rw - cen * p.P);
        gr_ = p.gval + cen * C;
        ar_ = p.amax + cen * C;
    };
    // the 16 channels a lane consumes next (upstream gradient, arg-max bytes), one chunk ahead
    float4 g[4];
    unsigned am[4];
    auto issue = [&](const float *gr_, const unsigned char *ar_, int ci) {
        const int k0 = ci * 32 + h * 16;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            g[q] = *(const float4 *)(gr_ + k0 + 4 * q);
            am[q] = *(const unsigned *)(ar_ + k0 + 4 * q);
        }
    };
    if ((long long)blockIdx.x * 4 + wave < ntile) {
        const float *g0;
        const unsigned char *a0;
        int p0;
        tileptrs((long long)blockIdx.x * 4 + wave, g0, a0, p0);
        issue(g0, a0, 0);
    }
    for (long long tile = (long long)blockIdx.x * 4 + wave; tile < ntile; tile += (long long)gridDim.x * 4) {
        const long long r0 = tile << 5;
        const int nrows = (p.E - r0 < 32) ? (int)(p.E - r0) : 32;
        const float *gr, *ngr;
        const unsigned char *ar, *nar;
        int pp, pn;
        tileptrs(tile, gr, ar, pp);
        ngr = gr; nar = ar;
        {
            const long long tn = tile + (long long)gridDim.x * 4;
            if (tn < ntile) tileptrs(tn, ngr, nar, pn);
        }
        // the layer in front, twice: C/D row order (rows (r&3) + 8(r>>2) + 4h, column l31) as the A operand of
        // the dW^T / S2 products and for the epilogue's sums; row order (lane = row, 16 consecutive columns) as
        // the A operand of the dense product.  NaN where the lane has no row: its activation is then 0.
        const long long base = (r0 + 4 * h) * GG_NZ_K + l31;
        float avr[16], zpv[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int rr = (r & 3) + 8 * (r >> 2);
            const bool ok = nrows == 32 || rr + 4 * h < nrows;
            const float v = *(ok ? p.Z1 + base + rr * GG_NZ_K : p.Z1);      // (no branch around the load)
            zpv[r] = ok ? v : __builtin_nanf("");
        }
        long long rw = r0 + l31;
        if (rw >= p.E) rw = p.E - 1;
        float4 z1r[4];
#pragma unroll
        for (int q = 0; q < 4; q++) z1r[q] = *(const float4 *)(p.Z1 + rw * GG_NZ_K + 16 * h + 4 * q);
        ggm_f32x16 accx;
#pragma unroll
        for (int r = 0; r < 16; r++) accx[r] = 0.f;
#pragma unroll
        for (int ci = 0; ci < NJ; ci++) {
            const int cc = ci & 1;
            float4 a[4];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const gg_f32x4 sg = __builtin_bit_cast(gg_f32x4, g[q]);      // (scale and ReLU mask applied upstream)
                gg_f32x4 d;
                d.x = (int)(am[q] & 255u) == pp ? sg.x : 0.f;
                d.y = (int)((am[q] >> 8) & 255u) == pp ? sg.y : 0.f;
                d.z = (int)((am[q] >> 16) & 255u) == pp ? sg.z : 0.f;
                d.w = (int)(am[q] >> 24) == pp ? sg.w : 0.f;
                a[q] = __builtin_bit_cast(float4, d);
                float *tw = T + (cc * 32 + h * 16 + 4 * q) * GG_NZ_TS + l31;
                tw[0] = d.x; tw[GG_NZ_TS] = d.y; tw[2 * GG_NZ_TS] = d.z; tw[3 * GG_NZ_TS] = d.w;
            }
            if (ci + 1 < NJ) issue(gr, ar, ci + 1);
            else issue(ngr, nar, 0);
            // (keep the loads HERE: left alone, the scheduler sinks them to their first use)
            __builtin_amdgcn_sched_barrier(0);
            {
                gg_f32x4 w4[4];
#pragma unroll
                for (int q = 0; q < 4; q++) w4[q] = gg_ld_f4(Wl + (ci * 64 + lane) * GG_NZ_WS + 4 * q);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float av[4] = {a[q].x, a[q].y, a[q].z, a[q].w};
                    const float wv[4] = {w4[q].x, w4[q].y, w4[q].z, w4[q].w};
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        accx = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], wv[i], accx, 0, 0, 0);
                }
            }
            // (the activations of the layer in front are formed HERE, behind the first chunk's dZ and dX MFMAs:
            //  at the top of the tile the wave would wait out the full latency of the loads it has just issued)
            if (ci == 0) {
#pragma unroll
                for (int r = 0; r < 16; r++) avr[r] = fmaxf(__builtin_fmaf(zpv[r], ps, psh), 0.f);   // 0 where zpv is NaN
            }
            // dW^T tile ci += a1^T dZ(channels cc*32.. of the LDS tile).  The tile belongs to this wave alone: its
            // LDS writes only have to land before its reads; the channels read here are overwritten two chunks
            // later, behind another of these barriers.
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            {
                gg_f32x4 t4[4];
#pragma unroll
                for (int jg = 0; jg < 4; jg++) t4[jg] = gg_ld_f4(T + (cc * 32 + l31) * GG_NZ_TS + 8 * jg + 4 * h);
#pragma unroll
                for (int jg = 0; jg < 4; jg++) {
                    const float tv[4] = {t4[jg].x, t4[jg].y, t4[jg].z, t4[jg].w};
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        accw[ci] = __builtin_amdgcn_mfma_f32_32x32x2f32(avr[4 * jg + i], tv[i], accw[ci], 0, 0, 0);
                }
            }
        }
        // dense term: a1 (row order) x M, and S2 += a1^T a1
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const gg_f32x4 y = gg_bnrelu4v(__builtin_bit_cast(gg_f32x4, z1r[q]), gg_ld_f4(pcs + 16 * h + 4 * q),
                                           gg_ld_f4(pcs + 32 + 16 * h + 4 * q));
            const float yv[4] = {y.x, y.y, y.z, y.w};
            const gg_f32x4 m4 = gg_ld_f4(Mp + lane * GG_NZ_WS + 4 * q);
            const float mv[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
            for (int i = 0; i < 4; i++) accx = __builtin_amdgcn_mfma_f32_32x32x2f32(yv[i], mv[i], accx, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) accS = __builtin_amdgcn_mfma_f32_32x32x2f32(avr[r], avr[r], accS, 0, 0, 0);
        // dX tile + BatchNorm-backward sums of the layer in front + S1
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
        {
            float *xp = p.dX + base;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rr = (r & 3) + 8 * (r >> 2);
                if (nrows == 32 || rr + 4 * h < nrows) {
                    const float dx = accx[r] + v0l;
                    xp[rr * GG_NZ_K] = dx;
                    const float d = avr[r] > 0.f ? dx : 0.f;
                    s1 += d;
                    s2 = __builtin_fmaf(d, __builtin_fmaf(zpv[r], pr, pc), s2);
                    s3 += avr[r];
                }
            }
        }
        a1 += s1;
        a2 += s2;
        a3 += s3;
    }
    // partial tiles: the four waves add up in LDS (fixed order), one [tile][reg][lane] block per workgroup
    {
        float *blk = pcs + 64;                         // 5 * 1024 floats over the tile area (4 * 64 * 36)
        __syncthreads();
        for (int w = 0; w < 4; w++) {
            if (wave == w) {
#pragma unroll
                for (int j = 0; j < NJ + 1; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int idx = (j * 16 + r) * 64 + lane;
                        blk[idx] = (w == 0 ? 0.f : blk[idx]) + (j < NJ ? accw[j < NJ ? j : 0][r] : accS[r]);
                    }
            }
            __syncthreads();
        }
        float *part = p.part + (size_t)blockIdx.x * (NJ + 1) * 1024;
        for (int i = tid; i < (NJ + 1) * 1024; i += 256) part[i] = blk[i];
    }
    __syncthreads();
    float *red = lds;                                  // [4 waves][3][32]
    {
        const float t1 = a1 + __shfl_xor(a1, 32, 64);
        const float t2 = a2 + __shfl_xor(a2, 32, 64);
        const float t3 = a3 + __shfl_xor(a3, 32, 64);
        if (lane < 32) {
            red[(wave * 3 + 0) * 32 + lane] = t1;
            red[(wave * 3 + 1) * 32 + lane] = t2;
            red[(wave * 3 + 2) * 32 + lane] = t3;
        }
    }
    __syncthreads();
    if (tid < 96) {
        const int which = tid >> 5, col = tid & 31;
        float v = 0.f;
        for (int w = 0; w < 4; w++) v += red[(w * 3 + which) * 32 + col];
        if (which < 2) atomicAdd(&p.psums[which * GG_NZ_K + col], (double)v);
        else atomicAdd(&p.s1[col], (double)v);
    }
}

// ------------------------------------------------------------------------------------------------------------
// gg_k_att_bwd_nz2 (round 5): the same arithmetic in the same order -- every output word equals gg_k_att_bwd_nz's
// -- with the tile loop stripped of what the ISA showed it spending outside its 160 MFMAs (tools/isa.py: 575 VALU +
// 330 SALU per tile): two 64-bit divisions per tile (row -> centre, neighbour), 16 predicated 64-bit addresses for
// the layer-in-front rows, 16 row branches in the epilogue, NaN selects.
//   * (centre, neighbour) of a lane's row advance by a constant per iteration: add, compare, select -- no division;
//   * every stream is a buffer access with a range-checked descriptor: rows past the end load 0 / are not stored,
//     so a partial last tile needs no predicate on any load or store (its idle rows get a1 = 0 under ONE
//     wave-uniform branch); gval / amax of a lane are 4 + 1 sixteen-byte loads per chunk with immediate offsets;
//   * the descriptor of the tile's Z1 / dX rows is rebased per tile on the scalar unit.
// Needs 32-bit byte offsets: E < 2^24 rows, E / P < 2^22 centres (gg_att_nz2_ok); otherwise the first form runs.
__device__ gg_i32x4 gg_buf_ld4i(gg_rsrc r, unsigned lane_bytes, unsigned uniform_bytes, int aux = 0) __asm("llvm.amdgcn.raw.buffer.load.v4i32");

//
// S2IN (round 6): the moments S1 = sum_e a1, S2 = sum_e a1 a1^T are NOT accumulated here -- the Z2-free FORWARD of the
// same layer has already formed them, in fp64, for the BatchNorm statistics of this conv (gg_k_att_moments,
// gridgcn_attfwd.hip), and the dense part of dW2 takes them from there (gg_k_att_nz_reduce_fin): 144 MFMAs per tile
// instead of 160, sixteen accumulator registers and the S1 column sums gone from the loop.  Every other word of the
// kernel is the <false> form.
template <bool S2IN>
__global__ __launch_bounds__(256, 2) void gg_k_att_bwd_nz2(GGAttNz p)
{
    constexpr int C = GG_NZ_C, NJ = 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    float *Wl = lds;                       // [4 chunks][64 lanes][WS]   dX operand
    float *Mp = Wl + 4 * 64 * GG_NZ_WS;    // [64 lanes][WS]             dense operand M
    float *csc = Mp + 64 * GG_NZ_WS;       // [C] scale
    float *cbz = csc + C;                  // [C] bz = -(sc rstd) m2
    float *ct = cbz + C;                   // [C] cz + bz (b2 - mu)
    float *v0 = ct + C;                    // [32]
    float *pcs = v0 + 32;                  // [2][32] previous layer's scale, shift
    float *T = pcs + 64 + wave * (64 * GG_NZ_TS);
    for (int i0 = tid; i0 < C * 32; i0 += 256 * 8) {
        float w8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) w8[u] = p.W2[i0 + 256 * u < C * 32 ? i0 + 256 * u : C * 32 - 1];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + 256 * u < C * 32 ? i0 + 256 * u : C * 32 - 1;
            const int k = i >> 5, col = i & 31;
            Wl[((k >> 5) * 64 + ((k >> 4) & 1) * 32 + col) * GG_NZ_WS + (k & 15)] = w8[u];
        }
    }
    for (int c = tid; c < C; c += 256) {
        const float sc = p.sc[c];
        const float m1 = (float)(p.bsums[c] / (double)p.E), m2 = (float)(p.bsums[C + c] / (double)p.E);
        const float bz = -(sc * p.rs[c]) * m2, cz = -(sc * m1);
        csc[c] = sc;
        cbz[c] = bz;
        ct[c] = cz + bz * (p.b2[c] - p.mu[c]);
    }
    if (tid < 32) { pcs[tid] = p.ps[tid]; pcs[32 + tid] = p.psh[tid]; }
    __syncthreads();
    for (int i = tid; i < 1024; i += 256) {
        const int s = i >> 6, ln = i & 63, j = 16 * (ln >> 5) + s, col = ln & 31;
        float m = 0.f;
        for (int c = 0; c < C; c++) m = __builtin_fmaf(cbz[c] * gg_nz_w2(Wl, c, j), gg_nz_w2(Wl, c, col), m);
        Mp[ln * GG_NZ_WS + s] = m;
    }
    if (tid < 32) {
        float v = 0.f;
        for (int c = 0; c < C; c++) v = __builtin_fmaf(ct[c], gg_nz_w2(Wl, c, tid), v);
        v0[tid] = v;
    }
    __syncthreads();
    const float ps = pcs[l31], psh = pcs[32 + l31];
    const float pm = p.pm[l31], pr = p.pr[l31];
    const float pc = -(pm * pr);                       // zhat = zp * pr + pc
    const float v0l = v0[l31];
    float a1 = 0.f, a2 = 0.f, a3 = 0.f;
    ggm_f32x16 accw[NJ], accS;
    ggm_zero<NJ>(accw);
    if constexpr (!S2IN) {
#pragma unroll
        for (int r = 0; r < 16; r++) accS[r] = 0.f;
    }

    const int E = (int)p.E, P = p.P;
    const int ntile = (E + 31) >> 5;
    const int tstride = (int)gridDim.x * 4;
    int tile = (int)blockIdx.x * 4 + wave;             // (wave uniform)
    // (centre, neighbour) of this lane's row, now and one iteration (tstride tiles) on: constants qs, rs
    const int qs = (tstride * 32) / P, rs = tstride * 32 - qs * P;
    int cen, pp;
    {
        const int row = tile * 32 + l31;
        cen = row / P;
        pp = row - cen * P;
    }
    const unsigned ncent = (unsigned)(E / P);
    const gg_rsrc rg = gg_make_rsrc_n(p.gval, ncent * (unsigned)(C * 4));
    const gg_rsrc ra = gg_make_rsrc_n(p.amax, ncent * (unsigned)C);
    // lane constants of the two row layouts of a 32-row block of [.][32] floats
    const unsigned vcd = (unsigned)(h * 4 * GG_NZ_K + l31) * 4u;           // C/D order: row 4h (+ rr), column l31
    const unsigned vrw = (unsigned)(l31 * GG_NZ_K + 16 * h) * 4u;          // row order: row l31, columns 16h ..

    // the 16 channels a lane consumes next (upstream gradient, arg-max bytes), one chunk ahead
    gg_f32x4 g[4];
    gg_i32x4 am;
    auto issue = [&](int cen_, int ci) {
        const unsigned vg = (unsigned)cen_ * (unsigned)(C * 4) + (unsigned)(ci * 32 + h * 16) * 4u;
        const unsigned va = (unsigned)cen_ * (unsigned)C + (unsigned)(ci * 32 + h * 16);
#pragma unroll
        for (int q = 0; q < 4; q++) g[q] = gg_buf_ld4(rg, vg + 16u * q, 0);
        am = gg_buf_ld4i(ra, va, 0);
    };
    if (tile < ntile) issue(cen, 0);
    for (; tile < ntile; tile += tstride) {
        const int r0 = tile << 5;
        const int nrows = E - r0 < 32 ? E - r0 : 32;                        // (wave uniform)
        int cenN = cen + qs, ppN = pp + rs;
        { const bool t = ppN >= P; ppN -= t ? P : 0; cenN += t ? 1 : 0; }
        const gg_rsrc rz = gg_make_rsrc_n(p.Z1 + (size_t)r0 * GG_NZ_K, (unsigned)nrows * (GG_NZ_K * 4));
        const gg_rsrc rx = gg_make_rsrc_n(p.dX + (size_t)r0 * GG_NZ_K, (unsigned)nrows * (GG_NZ_K * 4));
        // the layer in front, twice: C/D row order (rows (r&3) + 8(r>>2) + 4h, column l31) as the A operand of
        // the dW^T / S2 products and for the epilogue's sums; row order (lane = row, 16 consecutive columns) as
        // the A operand of the dense product.  Rows past the end load 0.
        float avr[16], zpv[16];
#pragma unroll
        for (int r = 0; r < 16; r++) zpv[r] = gg_buf_ld(rz, vcd + (unsigned)(((r & 3) + 8 * (r >> 2)) * GG_NZ_K * 4), 0);
        gg_f32x4 z1r[4];
#pragma unroll
        for (int q = 0; q < 4; q++) z1r[q] = gg_buf_ld4(rz, vrw + 16u * q, 0);
        ggm_f32x16 accx;
#pragma unroll
        for (int r = 0; r < 16; r++) accx[r] = 0.f;
#pragma unroll
        for (int ci = 0; ci < NJ; ci++) {
            const int cc = ci & 1;
            float a[4][4];
            __builtin_amdgcn_sched_barrier(0);
            {
                const unsigned amv[4] = {(unsigned)am.x, (unsigned)am.y, (unsigned)am.z, (unsigned)am.w};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    a[q][0] = (int)(amv[q] & 255u) == pp ? g[q].x : 0.f;     // (scale and ReLU mask applied upstream)
                    a[q][1] = (int)((amv[q] >> 8) & 255u) == pp ? g[q].y : 0.f;
                    a[q][2] = (int)((amv[q] >> 16) & 255u) == pp ? g[q].z : 0.f;
                    a[q][3] = (int)(amv[q] >> 24) == pp ? g[q].w : 0.f;
                    float *tw = T + (cc * 32 + h * 16 + 4 * q) * GG_NZ_TS + l31;
                    tw[0] = a[q][0]; tw[GG_NZ_TS] = a[q][1]; tw[2 * GG_NZ_TS] = a[q][2]; tw[3 * GG_NZ_TS] = a[q][3];
                }
            }
            if (ci + 1 < NJ) issue(cen, ci + 1);
            else issue(cenN, 0);
            // (keep the loads HERE: left alone, the scheduler sinks them to their first use)
            __builtin_amdgcn_sched_barrier(0);
            {
                gg_f32x4 w4[4];
#pragma unroll
                for (int q = 0; q < 4; q++) w4[q] = gg_ld_f4(Wl + (ci * 64 + lane) * GG_NZ_WS + 4 * q);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float wv[4] = {w4[q].x, w4[q].y, w4[q].z, w4[q].w};
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        accx = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][i], wv[i], accx, 0, 0, 0);
                }
            }
            // (the activations of the layer in front are formed HERE, behind the first chunk's dZ and dX MFMAs:
            //  at the top of the tile the wave would wait out the full latency of the loads it has just issued)
            if (ci == 0) {
#pragma unroll
                for (int r = 0; r < 16; r++) avr[r] = fmaxf(__builtin_fmaf(zpv[r], ps, psh), 0.f);
                if (nrows < 32) {                      // (wave uniform: the last tile only)
#pragma unroll
                    for (int r = 0; r < 16; r++) avr[r] = ((r & 3) + 8 * (r >> 2) + 4 * h < nrows) ? avr[r] : 0.f;
                }
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            {
                gg_f32x4 t4[4];
#pragma unroll
                for (int jg = 0; jg < 4; jg++) t4[jg] = gg_ld_f4(T + (cc * 32 + l31) * GG_NZ_TS + 8 * jg + 4 * h);
#pragma unroll
                for (int jg = 0; jg < 4; jg++) {
                    const float tv[4] = {t4[jg].x, t4[jg].y, t4[jg].z, t4[jg].w};
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        accw[ci] = __builtin_amdgcn_mfma_f32_32x32x2f32(avr[4 * jg + i], tv[i], accw[ci], 0, 0, 0);
                }
            }
        }
        // dense term: a1 (row order) x M, and S2 += a1^T a1
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const gg_f32x4 y = gg_bnrelu4v(z1r[q], gg_ld_f4(pcs + 16 * h + 4 * q), gg_ld_f4(pcs + 32 + 16 * h + 4 * q));
            const float yv[4] = {y.x, y.y, y.z, y.w};
            const gg_f32x4 m4 = gg_ld_f4(Mp + lane * GG_NZ_WS + 4 * q);
            const float mv[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
            for (int i = 0; i < 4; i++) accx = __builtin_amdgcn_mfma_f32_32x32x2f32(yv[i], mv[i], accx, 0, 0, 0);
        }
        if constexpr (!S2IN) {
#pragma unroll
            for (int r = 0; r < 16; r++) accS = __builtin_amdgcn_mfma_f32_32x32x2f32(avr[r], avr[r], accS, 0, 0, 0);
        }
        // dX tile + BatchNorm-backward sums of the layer in front + S1 (rows past the end: a1 = 0, store dropped)
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float dx = accx[r] + v0l;
            gg_buf_st(dx, rx, vcd + (unsigned)(((r & 3) + 8 * (r >> 2)) * GG_NZ_K * 4), 0);
            const float d = avr[r] > 0.f ? dx : 0.f;
            s1 += d;
            s2 = __builtin_fmaf(d, __builtin_fmaf(zpv[r], pr, pc), s2);
            if constexpr (!S2IN) s3 += avr[r];
        }
        a1 += s1;
        a2 += s2;
        if constexpr (!S2IN) a3 += s3;
        cen = cenN;
        pp = ppN;
    }
    // partial tiles: the four waves add up in LDS (fixed order), one [tile][reg][lane] block per workgroup
    // (S2IN: the four dW^T tiles only; the block stride of `part` stays five tiles)
    {
        constexpr int NPT = S2IN ? NJ : NJ + 1;
        float *blk = pcs + 64;                         // 5 * 1024 floats over the tile area (4 * 64 * 36)
        __syncthreads();
        for (int w = 0; w < 4; w++) {
            if (wave == w) {
#pragma unroll
                for (int j = 0; j < NPT; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int idx = (j * 16 + r) * 64 + lane;
                        blk[idx] = (w == 0 ? 0.f : blk[idx]) + (j < NJ ? accw[j < NJ ? j : 0][r] : accS[r]);
                    }
            }
            __syncthreads();
        }
        float *part = p.part + (size_t)blockIdx.x * (NJ + 1) * 1024;
        for (int i = tid; i < NPT * 1024; i += 256) part[i] = blk[i];
    }
    __syncthreads();
    float *red = lds;                                  // [4 waves][3][32]
    {
        const float t1 = a1 + __shfl_xor(a1, 32, 64);
        const float t2 = a2 + __shfl_xor(a2, 32, 64);
        const float t3 = a3 + __shfl_xor(a3, 32, 64);
        if (lane < 32) {
            red[(wave * 3 + 0) * 32 + lane] = t1;
            red[(wave * 3 + 1) * 32 + lane] = t2;
            red[(wave * 3 + 2) * 32 + lane] = t3;
        }
    }
    __syncthreads();
    if (tid < (S2IN ? 64 : 96)) {
        const int which = tid >> 5, col = tid & 31;
        float v = 0.f;
        for (int w = 0; w < 4; w++) v += red[(w * 3 + which) * 32 + col];
        if (which < 2) atomicAdd(&p.psums[which * GG_NZ_K + col], (double)v);
        else atomicAdd(&p.s1[col], (double)v);
    }
}

// Sum of the workgroups' partial tiles.  One 1024-thread workgroup per (tile j, register r): 16 groups of 64
// lanes each sum a slice of the workgroups, LDS adds the groups in a fixed order.  Tile j < 4, lane l, register r
// hold dW^T[i][ch], ch = 32j + (l & 31), i = (r & 3) + 8(r >> 2) + 4(l >> 5); tile 4 holds S2[i][l & 31].
__global__ __launch_bounds__(1024) void gg_k_att_nz_reduce(const float *__restrict__ part, int nwg,
                                                           float *__restrict__ dW, float *__restrict__ S2)
{
    __shared__ float sh[16][64];
    const int j = blockIdx.x >> 4, r = blockIdx.x & 15;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    float v = 0.f;
    int w = grp;
    for (; w + 7 * 16 < nwg; w += 8 * 16) {             // eight loads in flight, added in the same order as one by one
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) t[u] = part[((size_t)(w + 16 * u) * 5 + j) * 1024 + r * 64 + lane];
#pragma unroll
        for (int u = 0; u < 8; u++) v += t[u];
    }
    for (; w < nwg; w += 16) v += part[((size_t)w * 5 + j) * 1024 + r * 64 + lane];
    sh[grp][lane] = v;
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
        for (int g = 0; g < 16; g++) t += sh[g][lane];
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (j < 4) dW[(32 * j + (lane & 31)) * GG_NZ_K + i] = t;
        else S2[i * GG_NZ_K + (lane & 31)] = t;
    }
}

// dW[c][i] += (cz_c + bz_c (b2_c - mu_c)) S1[i] + bz_c sum_k W2[c][k] S2[k][i]; the layer's BatchNorm-backward
// vectors (m1, m2, dgamma, dbeta) from its sums, as the other backward kernels' reduce launches write them
__global__ __launch_bounds__(256) void gg_k_att_nz_finish(GGAttNz p, const float *__restrict__ S2, float *__restrict__ dW,
                                                          float *__restrict__ fm1, float *__restrict__ fm2,
                                                          float *__restrict__ fdg, float *__restrict__ fdb)
{
    constexpr int C = GG_NZ_C;
    const int t = blockIdx.x * 256 + threadIdx.x;     // 128 x 32
    const int c = t >> 5, i = t & 31;
    if (c >= C) return;
    const float sc = p.sc[c];
    const float m1 = (float)(p.bsums[c] / (double)p.E), m2 = (float)(p.bsums[C + c] / (double)p.E);
    const float bz = -(sc * p.rs[c]) * m2, cz = -(sc * m1);
    float acc = 0.f;
    for (int k = 0; k < GG_NZ_K; k++) acc = __builtin_fmaf(p.W2[c * GG_NZ_K + k], S2[k * GG_NZ_K + i], acc);
    dW[t] += (cz + bz * (p.b2[c] - p.mu[c])) * (float)p.s1[i] + bz * acc;
    if (i == 0 && fm1) gg_bn_bwd_fin_write(p.bsums, p.E, C, c, fm1, fm2, fdg, fdb);
}

// S2IN form: reduce + finish in ONE launch.  Block (j < 4, r) sums the workgroups' partial dW^T values of tile j,
// register r exactly as gg_k_att_nz_reduce does, and its 64 finishing threads add the dense part of their own element
//   dW[c][i] += (cz_c + bz_c (b2_c - mu_c)) S1[i] + bz_c sum_k W2[c][k] S2[k][i]
// from the forward's moments (fp64; mom[r][l] = S2[ggm_row(r, l)][l & 31], mom[1024 + l] = lane l's share of
// S1[l & 31]: gg_k_att_moments_reduce); block 0 also writes the layer's BatchNorm-backward vectors.
__global__ __launch_bounds__(1024) void gg_k_att_nz_reduce_fin(GGAttNz p, int nwg, const double *__restrict__ mom,
                                                               float *__restrict__ dW, float *__restrict__ fm1,
                                                               float *__restrict__ fm2, float *__restrict__ fdg,
                                                               float *__restrict__ fdb)
{
    constexpr int C = GG_NZ_C;
    __shared__ float sh[16][64];
    const int j = blockIdx.x >> 4, r = blockIdx.x & 15;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    float v = 0.f;
    int w = grp;
    for (; w + 7 * 16 < nwg; w += 8 * 16) {             // (the summation order of gg_k_att_nz_reduce)
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) t[u] = p.part[((size_t)(w + 16 * u) * 5 + j) * 1024 + r * 64 + lane];
#pragma unroll
        for (int u = 0; u < 8; u++) v += t[u];
    }
    for (; w < nwg; w += 16) v += p.part[((size_t)w * 5 + j) * 1024 + r * 64 + lane];
    sh[grp][lane] = v;
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
        for (int g = 0; g < 16; g++) t += sh[g][lane];
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int c = 32 * j + (lane & 31);
        const float sc = p.sc[c];
        const float m1 = (float)(p.bsums[c] / (double)p.E), m2 = (float)(p.bsums[C + c] / (double)p.E);
        const float bz = -(sc * p.rs[c]) * m2, cz = -(sc * m1);
        // S2[k][i]: lane i + 32 ((k >> 2) & 1), register (k & 3) + 4 (k >> 3)
        double acc = 0.0;
        for (int k = 0; k < GG_NZ_K; k++)
            acc += (double)p.W2[c * GG_NZ_K + k] * mom[((k & 3) + 4 * (k >> 3)) * 64 + i + 32 * ((k >> 2) & 1)];
        const double s1i = mom[1024 + i] + mom[1024 + 32 + i];
        dW[c * GG_NZ_K + i] = t + (float)((double)(cz + bz * (p.b2[c] - p.mu[c])) * s1i + (double)bz * acc);
    }
    if (blockIdx.x == 0 && threadIdx.x >= 64 && threadIdx.x < 64 + C && fm1)
        gg_bn_bwd_fin_write(p.bsums, p.E, C, (int)threadIdx.x - 64, fm1, fm2, fdg, fdb);
}

// GRIDGCN_OPT_ATT_NZ_V2 [1]: the stripped tile loop (identical results); 0 = the round-4 kernel
static int g_att_nz_v2 = 1;
void gg_set_att_nz_v2(int v) { g_att_nz_v2 = v ? 1 : 0; }
int gg_get_att_nz_v2() { return g_att_nz_v2; }
// 32-bit byte offsets into gval ([E / P][128] floats) and row numbers with room for one iteration's advance
static bool gg_att_nz2_ok(long long E, int P) { return E < (1ll << 24) && E / P < (1ll << 22); }

static int gg_att_nz_grid(long long E)
{
    const long long ntile = (E + 31) >> 5;
    long long nb = (ntile + 3) / 4;
    if (nb > 256 * 2) nb = 256 * 2;        // two resident workgroups per CU (registers)
    return (int)(nb < 1 ? 1 : nb);
}

bool gg_att_bwd_noz_ok(long long E, int cin, int C) { return cin == GG_NZ_K && C == GG_NZ_C && E >= 32; }
// the form that takes S1 / S2 from the forward's moments exists for the stripped tile loop only
bool gg_att_bwd_noz_mom_ok(long long E, int cin, int C, int P)
{
    return gg_att_bwd_noz_ok(E, cin, C) && P >= 1 && (E % P) == 0 && gg_att_nz2_ok(E, P);
}

size_t gg_att_bwd_noz_workspace(long long E)
{
    return ((size_t)gg_att_nz_grid(E) * 5 * 1024 + 1024) * sizeof(float);
}

// workspace: gg_att_bwd_noz_workspace(E) bytes; psums [2][32] and s1 [32] (fp64) zero on entry
int gg_att_bwd_noz(const float *Z1, const float *ps, const float *psh, const float *pm, const float *pr,
                   const float *W2, const float *b2, const float *sc, const float *mu, const float *rs,
                   const double *bsums, const unsigned char *amax, const float *gval, int P, long long E,
                   float *dX, float *dW, float *m1, float *m2, float *dgamma, float *dbeta, double *psums,
                   double *s1, void *ws, hipStream_t st, const double *mom)
{
    if (E < 32 || P < 1 || (E % P)) return 1;
    static GGDevOnce attr_done;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)gg_k_att_bwd_nz, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) return 3;
        if (hipFuncSetAttribute((const void *)gg_k_att_bwd_nz2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) return 3;
        if (hipFuncSetAttribute((const void *)gg_k_att_bwd_nz2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) return 3;
        attr_done = true;
    }
    GGAttNz p;
    p.Z1 = Z1; p.ps = ps; p.psh = psh; p.pm = pm; p.pr = pr; p.W2 = W2; p.b2 = b2; p.sc = sc; p.mu = mu; p.rs = rs;
    p.bsums = bsums; p.amax = amax; p.gval = gval; p.dX = dX; p.part = (float *)ws; p.psums = psums; p.s1 = s1;
    p.E = E; p.P = P;
    const int nwg = gg_att_nz_grid(E);
    float *S2 = p.part + (size_t)nwg * 5 * 1024;
    const size_t lds = (size_t)(5 * 64 * GG_NZ_WS + 3 * GG_NZ_C + 32 + 64 + 4 * 64 * GG_NZ_TS) * sizeof(float);
    if (mom) {
        // the forward's moments stand in for S1 / S2: no accumulation of them in the tile loop, reduce + finish in one
        // launch (the stripped tile loop only: callers check gg_att_bwd_noz_mom_ok)
        if (!gg_att_nz2_ok(E, P)) return 1;
        gg_k_att_bwd_nz2<true><<<nwg, 256, lds, st>>>(p);
        gg_k_att_nz_reduce_fin<<<4 * 16, 1024, 0, st>>>(p, nwg, mom, dW, m1, m2, dgamma, dbeta);
        return hipGetLastError() == hipSuccess ? 0 : 3;
    }
    if (g_att_nz_v2 && gg_att_nz2_ok(E, P)) gg_k_att_bwd_nz2<false><<<nwg, 256, lds, st>>>(p);
    else gg_k_att_bwd_nz<<<nwg, 256, lds, st>>>(p);
    gg_k_att_nz_reduce<<<5 * 16, 1024, 0, st>>>(p.part, nwg, dW, S2);
    gg_k_att_nz_finish<<<(GG_NZ_C * GG_NZ_K + 255) / 256, 256, 0, st>>>(p, S2, dW, m1, m2, dgamma, dbeta);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
