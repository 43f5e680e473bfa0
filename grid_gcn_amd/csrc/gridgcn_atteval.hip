// gridgcn_atteval.hip -- evaluation-mode tail of a GridConv edge block in ONE kernel (gfx950):
//
//   agg[o, c] = max_p  relu(bn_p(Ysrc[src(e)][c] + Wg[:,c].geo(e) + b[c]))            point branch
//                    * relu(bn_a(W2[c,:] . relu(bn_1(Z1[e,:])) + b2[c]))               attention branch
//
// for the edges e = o*P + p of centre o (gcn_module_g_att.py:135-167 pair product, :57-59 max).  In
// evaluation every BatchNorm is a fixed affine map, so the second attention conv (K = 32 -> C), its
// activation, the product and the max need no pass over a materialised [E, C] tensor: the training
// path writes that tensor (1.7 GB at cfg4 up2) and reads it back in the max kernel.
//
// The conv runs as the TRANSPOSED product Z^T = W2 . h^T on v_mfma_f32_32x32x2_f32, which puts an
// edge in a lane and its channels in that lane's registers (lane l: edge l&31 of the tile; register r
// of tile t: channel 32t + (r&3) + 8(r>>2) + 4(l>>5)):
//   A operand  lane l holds W2[32t + (l&31)][k],  B operand  lane l holds h[edge l&31][k],
//   k = 16(l>>5) + s at step s -- the contraction order is free, so a lane reads the 16 consecutive
//   floats of ITS half of the edge's Z1 row with four 16-byte loads.
// A lane owns a CENTRE: the wave walks the P neighbours of its 32 centres one after the other (the
// p-th edges of 32 centres form one MFMA tile), folds attention x point activation into a running
// maximum in registers and writes the centre's row at the end -- no cross-lane step at all.  The
// point value of an (edge, channel quad) is a 16-byte gather from Ysrc.
#include "gridgcn_mma.h"
#include "gridgcn_atteval.h"


template <int NJ>
__global__ __launch_bounds__(256) void gg_k_att_max_eval(GGAttEval p)
{
    constexpr int C = NJ * 32;
    __shared__ __attribute__((aligned(16))) float Wz[16 * 64 * NJ];   // [step][lane][tile]
    __shared__ __attribute__((aligned(16))) float cst[8 * C];        // sa, ha', w0, w1, w2, bp, sp, hp
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    for (int i0 = tid; i0 < 16 * 64 * NJ; i0 += 256 * 8) {        // (eight loads in flight)
        float w8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + 256 * u < 16 * 64 * NJ ? i0 + 256 * u : 16 * 64 * NJ - 1;
            const int t = i % NJ, ln = (i / NJ) & 63, s = i / (NJ * 64);
            w8[u] = p.W2[(32 * t + (ln & 31)) * 32 + 16 * (ln >> 5) + s];
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            Wz[i0 + 256 * u < 16 * 64 * NJ ? i0 + 256 * u : 16 * 64 * NJ - 1] = w8[u];      // (unconditional store)
    }
    for (int c = tid; c < C; c += 256) {
        const float sa = p.sa[c];
        cst[c] = sa;
        cst[C + c] = p.b2[c] * sa + p.ha[c];           // conv bias folded into the shift
        cst[2 * C + c] = p.Wg ? p.Wg[c] : 0.f;
        cst[3 * C + c] = p.Wg ? p.Wg[C + c] : 0.f;
        cst[4 * C + c] = p.Wg ? p.Wg[2 * C + c] : 0.f;
        cst[5 * C + c] = p.bp[c];
        cst[6 * C + c] = p.sp[c];
        cst[7 * C + c] = p.hp[c];
    }
    __syncthreads();
    const int P = p.P;
    const long long ncent = p.E / P;
    const long long ntile = (ncent + 31) >> 5;         // 32 centres per tile: one centre per lane
    const long long rows = (long long)p.B * p.Nsrc;
    // previous BatchNorm for this lane's 16 k's
    float4 s1v[4], h1v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        s1v[q] = *(const float4 *)(p.s1 + 16 * h + 4 * q);
        h1v[q] = *(const float4 *)(p.h1 + 16 * h + 4 * q);
    }
    for (long long tile = (long long)blockIdx.x * 4 + wave; tile < ntile;
         tile += (long long)gridDim.x * 4) {
        long long o = (tile << 5) + l31;
        const bool live = o < ncent;
        if (!live) o = ncent - 1;
        const int bi = (int)(o / p.O);
        // running maximum of this centre's channels (the lane's half: 16 per tile)
        ggm_f32x16 best[NJ];
#pragma unroll
        for (int t = 0; t < NJ; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) best[t][r] = -__builtin_inff();
        for (int pp = 0; pp < P; pp++) {
            // (keeps the per-channel constants of the epilogue in LDS: hoisted out of this loop they
            // would occupy 512 registers)
            asm volatile("" ::: "memory");
            const long long e = o * P + pp;
            // h = relu(bn1(Z1[e][16h .. 16h+15]))
            float hv[16];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 z = *(const float4 *)(p.Z1 + e * 32 + 16 * h + 4 * q);
                hv[4 * q + 0] = fmaxf(z.x * s1v[q].x + h1v[q].x, 0.f);
                hv[4 * q + 1] = fmaxf(z.y * s1v[q].y + h1v[q].y, 0.f);
                hv[4 * q + 2] = fmaxf(z.z * s1v[q].z + h1v[q].z, 0.f);
                hv[4 * q + 3] = fmaxf(z.w * s1v[q].w + h1v[q].w, 0.f);
            }
            long long flat = (long long)p.nebidx[e] + (long long)bi * p.Nsrc;
            flat = flat < 0 ? 0 : (flat > rows - 1 ? rows - 1 : flat);
            const float4 ge = *(const float4 *)(p.att16 + e * 16);    // (dist, gx, gy, gz)
            const float *yrow = p.Ysrc + flat * C;
            // the neighbour's source row (this lane's 4 x NJ pieces) is requested BEFORE the MFMAs:
            // loaded piece by piece in the epilogue, each piece was a full L2 round trip in front of
            // 16 results (4 x NJ round trips per neighbour against 16 x NJ MFMAs)
            float4 yq[NJ * 4];
#pragma unroll
            for (int t = 0; t < NJ; t++)
#pragma unroll
                for (int qq = 0; qq < 4; qq++) yq[t * 4 + qq] = *(const float4 *)(yrow + 32 * t + 8 * qq + 4 * h);
            ggm_f32x16 acc[NJ];
            ggm_zero<NJ>(acc);
#pragma unroll
            for (int s = 0; s < 16; s++) {
                float w[NJ];
                if constexpr (NJ == 4) {
                    const float4 t = *(const float4 *)(Wz + (s * 64 + lane) * 4);
                    w[0] = t.x; w[1] = t.y; w[2] = t.z; w[3] = t.w;
                } else {
                    const float2 t = *(const float2 *)(Wz + (s * 64 + lane) * 2);
                    w[0] = t.x; w[1] = t.y;
                }
#pragma unroll
                for (int t = 0; t < NJ; t++)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t], hv[s], acc[t], 0, 0, 0);
            }
            // attention activation x point activation, folded into the running maximum
#pragma unroll
            for (int t = 0; t < NJ; t++) {
#pragma unroll
                for (int qq = 0; qq < 4; qq++) {
                    const int c = 32 * t + 8 * qq + 4 * h;
                    const float4 sa = *(const float4 *)(cst + c), ha = *(const float4 *)(cst + C + c);
                    const float4 w0 = *(const float4 *)(cst + 2 * C + c), w1 = *(const float4 *)(cst + 3 * C + c);
                    const float4 w2 = *(const float4 *)(cst + 4 * C + c), bp = *(const float4 *)(cst + 5 * C + c);
                    const float4 sp = *(const float4 *)(cst + 6 * C + c), hp = *(const float4 *)(cst + 7 * C + c);
                    const float4 y = yq[t * 4 + qq];
                    const float sav[4] = {sa.x, sa.y, sa.z, sa.w}, hav[4] = {ha.x, ha.y, ha.z, ha.w};
                    const float w0v[4] = {w0.x, w0.y, w0.z, w0.w}, w1v[4] = {w1.x, w1.y, w1.z, w1.w};
                    const float w2v[4] = {w2.x, w2.y, w2.z, w2.w}, bpv[4] = {bp.x, bp.y, bp.z, bp.w};
                    const float spv[4] = {sp.x, sp.y, sp.z, sp.w}, hpv[4] = {hp.x, hp.y, hp.z, hp.w};
                    const float yv[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const float y2 = fmaxf(acc[t][4 * qq + i] * sav[i] + hav[i], 0.f);
                        float z1 = yv[i];
                        z1 = fmaf(ge.y, w0v[i], z1);
                        z1 = fmaf(ge.z, w1v[i], z1);
                        z1 = fmaf(ge.w, w2v[i], z1);
                        z1 += bpv[i];
                        const float y1 = fmaxf(z1 * spv[i] + hpv[i], 0.f);
                        best[t][4 * qq + i] = fmaxf(best[t][4 * qq + i], y1 * y2);
                    }
                }
            }
        }
        if (live) {
#pragma unroll
            for (int t = 0; t < NJ; t++)
#pragma unroll
                for (int qq = 0; qq < 4; qq++)
                    *(float4 *)(p.out + o * p.ldo + 32 * t + 8 * qq + 4 * h) =
                        make_float4(best[t][4 * qq], best[t][4 * qq + 1], best[t][4 * qq + 2],
                                    best[t][4 * qq + 3]);
        }
    }
}

int gg_att_max_eval(const GGAttEval &p, int C, hipStream_t st)
{
    if ((C != 64 && C != 128) || p.P < 1 || p.E < 1 || (p.E % p.P) || (p.ldo & 3)) return 1;
    const long long ntile = (p.E / p.P + 31) >> 5;
    long long nb = (ntile + 3) / 4;
    if (nb > 256 * 2) nb = 256 * 2;   // resident workgroups per CU at 216 / 236 registers
    if (C == 64) gg_k_att_max_eval<2><<<(int)nb, 256, 0, st>>>(p);
    else gg_k_att_max_eval<4><<<(int)nb, 256, 0, st>>>(p);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
