// gridgcn_attfwd_nz.hip -- training forward of an up layer's attention branch WITHOUT the [E, 128] tensor
// (gfx950, fp32 MFMA).  Companion of gridgcn_attbwd_nz.hip.
//
// The second attention conv (update_att_mlp2d_scnd, gcn_module_g_att.py:152: 32 -> 128 channels per edge) used to
// write its pre-activation Z2 -- 1.68 GB at cfg4 up2 -- for two readers: the max kernel and the backward.  The
// backward no longer needs it (gridgcn_attbwd_nz.hip); the forward does not either:
//
//   * the BatchNorm statistics of z2 = W2 a1 + b2 are functions of the MOMENTS of a1 = relu(bn1(Z1)):
//       sum_e z2_c   = w_c . S1 + E b_c,            S1 = sum_e a1[e, :]            [32]
//       sum_e z2_c^2 = w_c^T S2 w_c + 2 b_c w_c . S1 + E b_c^2,   S2 = sum_e a1 a1^T   [32 x 32]
//     gg_k_att_moments: one pass over Z1 [E, 32] (a quarter of Z2), S2 by MFMA (a1^T a1 per 32-row tile);
//     gg_k_att_moments_fin: the two sums per channel in fp64 -> the ordinary gridgcn_bn_finalize.
//   * gg_k_att_pairmax: the conv, both activations, the product with the point branch and the max over the P
//     neighbours in one kernel.  The conv runs as Z2[(o, p), c] tiles "p-th edges of 32 centres" x 32 channels:
//     in the MFMA C/D layout a lane then owns ONE CHANNEL (every per-channel constant is a per-lane scalar) and
//     its 16 registers are 16 centres, so the running maximum over p, the arg max and the two pre-activations
//     kept for the backward live in registers -- no Z2 tile in LDS, no cross-lane step.  A wave owns one
//     32-channel tile of the layer and walks groups of 32 centres; the four waves of a workgroup share a group
//     (the rows of Z1 and of Ysrc they read are the same, from L1 / L2).
// Same operation order per element as the kernels it replaces (gg_k_linear_fwd_direct's k pairing, the point
// value's FMA chain of gg_k_pairmax_fwd4_src), so the pre-activations are the same bits; the BatchNorm vectors
// of the layer differ by the rounding of their sums (moments instead of a pass over Z2).
#include "gridgcn_mma.h"
#include "gridgcn_train.h"

#define GG_PM_C 128
#define GG_PM_K 32

// ------------------------------------------------------------------------------------------------------------
struct GGAttMom {
    const float *Z1;            // [E][32]
    const float *ps, *psh;      // [32] BatchNorm scale / shift of that layer
    float *part;                // [workgroups][1024] partial S2 tiles
    double *s1;                 // [32], zero on entry
    long long E;
};

__global__ __launch_bounds__(256) void gg_k_att_moments(GGAttMom p)
{
    __shared__ float blk[1024];
    __shared__ float red[4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const float ps = p.ps[l31], psh = p.psh[l31];
    ggm_f32x16 accS;
#pragma unroll
    for (int r = 0; r < 16; r++) accS[r] = 0.f;
    float a3 = 0.f;
    const long long ntile = (p.E + 31) >> 5;
    for (long long tile = (long long)blockIdx.x * 4 + wave; tile < ntile; tile += (long long)gridDim.x * 4) {
        const long long r0 = tile << 5;
        const int nrows = (p.E - r0 < 32) ? (int)(p.E - r0) : 32;
        const long long base = (r0 + 4 * h) * GG_PM_K + l31;
        float avr[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int rr = (r & 3) + 8 * (r >> 2);
            const bool ok = nrows == 32 || rr + 4 * h < nrows;
            const float v = *(ok ? p.Z1 + base + rr * GG_PM_K : p.Z1);       // (no branch around the load)
            avr[r] = ok ? fmaxf(__builtin_fmaf(v, ps, psh), 0.f) : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            accS = __builtin_amdgcn_mfma_f32_32x32x2f32(avr[r], avr[r], accS, 0, 0, 0);
            a3 += avr[r];
        }
    }
    // the four waves add up in LDS (fixed order): one [reg][lane] block per workgroup
    for (int w = 0; w < 4; w++) {
        if (wave == w) {
#pragma unroll
            for (int r = 0; r < 16; r++) blk[r * 64 + lane] = (w == 0 ? 0.f : blk[r * 64 + lane]) + accS[r];
        }
        __syncthreads();
    }
    for (int i = tid; i < 1024; i += 256) p.part[(size_t)blockIdx.x * 1024 + i] = blk[i];
    {
        const float t3 = a3 + __shfl_xor(a3, 32, 64);
        if (lane < 32) red[wave][lane] = t3;
    }
    __syncthreads();
    if (tid < 32) atomicAdd(&p.s1[tid], (double)((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])));
}

// S2 = sum of the partial tiles (register r, lane l of a tile hold S2[(r&3) + 8(r>>2) + 4(l>>5)][l & 31]); then
// per channel c: sums[c] = sum_e z2_c, sums[C + c] = sum_e z2_c^2 in fp64 -- what gg_k_bn_finalize expects
__global__ __launch_bounds__(1024) void gg_k_att_moments_fin(const float *__restrict__ part, int nwg,
                                                             const double *__restrict__ s1,
                                                             const float *__restrict__ W2,
                                                             const float *__restrict__ b2, long long E,
                                                             double *__restrict__ sums)
{
    __shared__ double S2[32][33];
    const int tid = threadIdx.x;
    {
        const int r = tid >> 6, lane = tid & 63;
        double v = 0.0;
        for (int w = 0; w < nwg; w++) v += (double)part[(size_t)w * 1024 + r * 64 + lane];
        S2[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][lane & 31] = v;
    }
    __syncthreads();
    if (tid < GG_PM_C) {
        const int c = tid;
        double m = 0.0, q = 0.0;
        for (int i = 0; i < GG_PM_K; i++) {
            const double wi = (double)W2[c * GG_PM_K + i];
            m += wi * s1[i];
            double t = 0.0;
            for (int j = 0; j < GG_PM_K; j++) t += S2[i][j] * (double)W2[c * GG_PM_K + j];
            q += wi * t;
        }
        const double b = (double)b2[c];
        sums[c] = m + (double)E * b;
        sums[GG_PM_C + c] = q + 2.0 * b * m + (double)E * b * b;
    }
}

// ------------------------------------------------------------------------------------------------------------
struct GGAttPm {
    const float *Z1;                    // [E][32]
    const float *ps, *psh;              // [32]
    const float *W2, *b2;               // [128][32], [128]
    const float *sa, *ha;               // [128] BatchNorm scale / shift of the attention layer
    const float *Ysrc;                  // [B * Nsrc][128] first point conv on the source points (nullptr: none)
    const int *nebidx;                  // [ncent * P]
    const float *att16;                 // [E][16]: (dist, gx, gy, gz, ...)
    const float *Wg;                    // [3][128] geo_vec weights of the point conv (nullptr: none)
    const float *bp;                    // [128] its bias
    const float *sp, *hp;               // [128] BatchNorm scale / shift of the point layer
    float *agg;                         // [ncent][lda]
    unsigned char *amax;                // [ncent][128]
    float *zsel;                        // [2][ncent][128]: point / attention pre-activation at the arg max
    long long ncent;
    int lda, Nsrc, O, B;
};

template <int PF>
__global__ __launch_bounds__(256, 2) void gg_k_att_pairmax(GGAttPm p)
{
    constexpr int C = GG_PM_C;
    __shared__ __attribute__((aligned(16))) float Gs[4][PF][32][4];   // per wave: (gx, gy, gz, source row) of (centre, p)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int c = 32 * wave + l31;                                    // this lane's channel in the C/D layout
    // B operand of the conv: lane (h, l31) holds W2[c][16 h + s], s < 16 -- constant over the kernel
    float wreg[16];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float4 t = *(const float4 *)(p.W2 + c * GG_PM_K + 16 * h + 4 * q);
        wreg[4 * q] = t.x; wreg[4 * q + 1] = t.y; wreg[4 * q + 2] = t.z; wreg[4 * q + 3] = t.w;
    }
    // previous BatchNorm for this lane's 16 k's of the A operand (row layout: lane = centre)
    float4 s1v[4], h1v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        s1v[q] = *(const float4 *)(p.ps + 16 * h + 4 * q);
        h1v[q] = *(const float4 *)(p.psh + 16 * h + 4 * q);
    }
    // per-channel constants = per-lane scalars
    const float b2c = p.b2[c], sac = p.sa[c], hac = p.ha[c];
    const float w0 = p.Wg ? p.Wg[c] : 0.f, w1 = p.Wg ? p.Wg[C + c] : 0.f, w2 = p.Wg ? p.Wg[2 * C + c] : 0.f;
    const float bpc = p.bp[c], spc = p.sp[c], hpc = p.hp[c];
    const bool hasY = p.Ysrc != nullptr;
    const float *ysrc = hasY ? p.Ysrc + c : p.bp;                    // (no source term: any valid address)
    const long long ystride = hasY ? C : 0;
    const long long rows = (long long)p.B * p.Nsrc;
    const long long ngroup = (p.ncent + 31) >> 5;
    for (long long grp = blockIdx.x; grp < ngroup; grp += gridDim.x) {
        const long long o0 = grp << 5;
        // ---- stage A: (geo_vec, source row) of every (centre, p) of the group; lane = centre ----
        long long oc = o0 + l31;
        if (oc >= p.ncent) oc = p.ncent - 1;
        {
            int nb[PF];
            float4 ge[PF];
#pragma unroll
            for (int pp = 0; pp < PF; pp++) {
                nb[pp] = p.nebidx[oc * PF + pp];
                ge[pp] = *(const float4 *)(p.att16 + (oc * PF + pp) * 16);
            }
            const long long bi = oc / p.O;
#pragma unroll
            for (int pp = 0; pp < PF; pp++) {
                long long flat = (long long)nb[pp] + bi * p.Nsrc;
                flat = flat < 0 ? 0 : (flat > rows - 1 ? rows - 1 : flat);
                if (h == 0)
                    *(float4 *)&Gs[wave][pp][l31][0] = make_float4(ge[pp].y, ge[pp].z, ge[pp].w, __int_as_float((int)flat));
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float best[16], zps[16], zas[16];
        int bix[16];
#pragma unroll
        for (int r = 0; r < 16; r++) { best[r] = -__builtin_inff(); zps[r] = 0.f; zas[r] = 0.f; bix[r] = 0; }
        // ---- stage B: neighbour by neighbour; the loads of neighbour p + 1 are issued before p is consumed ----
        float4 z1[2][4];
        float ys[2][16];
        auto issue = [&](int pp, int buf) {
#pragma unroll
            for (int q = 0; q < 4; q++) z1[buf][q] = *(const float4 *)(p.Z1 + (oc * PF + pp) * GG_PM_K + 16 * h + 4 * q);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int fl = __float_as_int(Gs[wave][pp][(r & 3) + 8 * (r >> 2) + 4 * h][3]);
                ys[buf][r] = ysrc[(long long)fl * ystride];
            }
        };
        issue(0, 0);
#pragma unroll
        for (int pp = 0; pp < PF; pp++) {
            const int buf = pp & 1;
            if (pp + 1 < PF) issue(pp + 1, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            // conv: Z2[(o, pp), c] = sum_k a1[(o, pp), k] W2[c, k]
            ggm_f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const gg_f32x4 y = gg_bnrelu4v(__builtin_bit_cast(gg_f32x4, z1[buf][q]), __builtin_bit_cast(gg_f32x4, s1v[q]),
                                               __builtin_bit_cast(gg_f32x4, h1v[q]));
                const float yv[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
                for (int i = 0; i < 4; i++)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(yv[i], wreg[4 * q + i], acc, 0, 0, 0);
            }
            // product with the point branch, running maximum over the neighbours (first maximum wins)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float z2 = acc[r] + b2c;
                const float y2 = fmaxf(z2 * sac + hac, 0.f);
                const float4 ge = *(const float4 *)&Gs[wave][pp][(r & 3) + 8 * (r >> 2) + 4 * h][0];
                float z0 = hasY ? ys[buf][r] : 0.f;
                z0 = fmaf(ge.x, w0, z0);
                z0 = fmaf(ge.y, w1, z0);
                z0 = fmaf(ge.z, w2, z0);
                z0 += bpc;
                const float y1 = fmaxf(z0 * spc + hpc, 0.f);
                const float v = y1 * y2;
                const bool upd = v > best[r];
                if (upd || pp == 0) { zps[r] = z0; zas[r] = z2; }
                if (upd) { best[r] = v; bix[r] = pp; }
            }
        }
        // ---- outputs: rows (r&3) + 8(r>>2) + 4h of the group, column c ----
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const long long o = o0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (o < p.ncent) {
                p.agg[o * p.lda + c] = best[r];
                p.amax[o * C + c] = (unsigned char)bix[r];
                p.zsel[o * C + c] = zps[r];
                p.zsel[p.ncent * C + o * C + c] = zas[r];
            }
        }
        __builtin_amdgcn_wave_barrier();        // (Gs is rewritten by the next group)
    }
}

size_t gg_att_moments_workspace(long long E)
{
    const long long ntile = (E + 31) >> 5;
    long long nb = (ntile + 3) / 4;
    if (nb > 512) nb = 512;
    return (size_t)(nb < 1 ? 1 : nb) * 1024 * sizeof(float);
}

// sums [2][128] fp64 (written); s1 [32] fp64 zero on entry; ws: gg_att_moments_workspace(E)
int gg_att_moments(const float *Z1, const float *ps, const float *psh, const float *W2, const float *b2, long long E,
                   double *sums, double *s1, void *ws, hipStream_t st)
{
    if (E < 1) return 1;
    GGAttMom p;
    p.Z1 = Z1; p.ps = ps; p.psh = psh; p.part = (float *)ws; p.s1 = s1; p.E = E;
    const int nwg = (int)(gg_att_moments_workspace(E) / (1024 * sizeof(float)));
    gg_k_att_moments<<<nwg, 256, 0, st>>>(p);
    gg_k_att_moments_fin<<<1, 1024, 0, st>>>(p.part, nwg, s1, W2, b2, E, sums);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

bool gg_att_pairmax_ok(int P, int cin, int C) { return P == 5 && cin == GG_PM_K && C == GG_PM_C; }

int gg_att_pairmax(const GGAttPm &p, int P, hipStream_t st)
{
    if (P != 5 || p.ncent < 1) return 1;
    const long long ngroup = (p.ncent + 31) >> 5;
    long long nb = ngroup < 256 * 8 ? ngroup : 256 * 8;
    gg_k_att_pairmax<5><<<(int)nb, 256, 0, st>>>(p);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

// flat-argument form for the C ABI (gridgcn_capi.hip)
int gg_att_pairmax_c(const float *Z1, const float *ps, const float *psh, const float *W2, const float *b2,
                     const float *sa, const float *ha, const float *Ysrc, const int *nebidx, const float *att16,
                     const float *Wg, const float *bp, const float *sp, const float *hp, int B, int Nsrc, int O,
                     int P, float *agg, int lda, unsigned char *amax, float *zsel, hipStream_t st)
{
    GGAttPm p;
    p.Z1 = Z1; p.ps = ps; p.psh = psh; p.W2 = W2; p.b2 = b2; p.sa = sa; p.ha = ha; p.Ysrc = Ysrc; p.nebidx = nebidx;
    p.att16 = att16; p.Wg = Wg; p.bp = bp; p.sp = sp; p.hp = hp; p.agg = agg; p.amax = amax; p.zsel = zsel;
    p.ncent = (long long)B * O; p.lda = lda; p.Nsrc = Nsrc; p.O = O; p.B = B;
    return gg_att_pairmax(p, P, st);
}
