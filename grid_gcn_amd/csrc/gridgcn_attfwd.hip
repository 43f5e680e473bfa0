// gridgcn_attfwd.hip -- TRAINING forward of the attention pair product / neighbour max of an up layer WITHOUT the
// [E, 128] pre-activation of the second attention conv (gfx950, wave64).
//
//   agg[o, c] = max_p  relu(bn_p(Ysrc[src(e)][c] + Wg[:,c].geo(e) + b[c]))          point branch (recomputed Z0)
//                    * relu(bn_2(W2[c,:] . relu(bn_1(Z1[e,:])) + b2[c]))              attention branch
//   e = 5 o + p  (gcn_module_g_att.py:135-167 update_att_mlp2d_scnd and the pair product, :57-59 the max pool)
//
// Until round 5 the 32 -> 128 conv wrote Z2 [E, 128] (1.68 GB at cfg4's up layer) only for the max kernel to read
// it once; the backward had stopped needing it in round 4 (gridgcn_attbwd_nz.hip).  Two things made the tensor
// necessary in training -- and both go away:
//
//   * its BatchNorm statistics.  z2 = W2 a1 + b2 is AFFINE in a1 = relu(bn_1(Z1)), so
//       sum_e z2[c]   = W2[c,:] . S1 + E b2[c]                         S1 = sum_e a1      [32]
//       sum_e z2[c]^2 = W2[c,:] S2 W2[c,:]^T + 2 b2[c] W2[c,:] . S1 + E b2[c]^2    S2 = sum_e a1 a1^T  [32, 32]
//     gg_k_att_moments reads Z1 once (420 MB instead of 1.68 GB written + read) and forms S2 on the MFMA unit
//     (a1^T a1: A and B operand are THE SAME register -- lane l holds a1[row0 + (l>>5)][l&31]); per-wave fp32
//     accumulators are folded into fp64 every 256 rows, the partials are summed in fp64 in a fixed order, the
//     quadratic form is evaluated in fp64.
//
//   * the product / max.  gg_k_att_pairmax runs the conv as  Z2 tile = a1 tile [32 rows, 32] x W2^T [32, 128]  with
//     an edge per A ROW and -- this is the point of the layout -- the 32 A rows of a tile chosen so that the C/D
//     registers of a lane hold WHOLE centres: lane (j, h) keeps rows (r&3) + 8(r>>2) + 4h, r = 0..15, of column
//     j, so A row i is given edge  30 t + 15 h_i + s_i  (h_i = (i>>2)&1, s_i = (i&3) + 4(i>>3); slot 15 is a
//     dummy): a tile is 6 centres = 30 consecutive edges, lane half h owns centres 6t + 3h .. + 2, five
//     registers each.  The max over a centre's neighbours is then a loop over registers: no cross-lane step, no
//     LDS transpose, and the point-branch gather Ysrc[src(e)][c] is ONE 128-byte coalesced load per half-wave
//     (lane = channel).  (The evaluation kernel gridgcn_atteval.hip uses the transposed product -- a lane owns
//     an edge -- whose gathers are 64 separate 16-byte pieces per load: 1.1 ms on the same layer.)
//     6.25 % of the MFMA rows are the dummies.
//
// Algorithmic bytes per edge: Z1 128 B (twice: moments, max), att16 16 B, index 4 B, the gathered Ysrc row through
// L2; per centre 128 x (4 + 1 + 8) B of outputs (agg, amax, zsel).
#include "gridgcn_mma.h"
#include "gridgcn_train.h"

struct GGAttFwd {
    const float *Ysrc;    // [B*Nsrc][128]
    const int *nebidx;    // [ncent*5]
    const float *att16;   // [ncent*5][16], geo_vec at 1..3
    const float *Wg;      // [3][128] or nullptr
    const float *b;       // [128]
    int Nsrc, O, B;
    const float *Z1;      // [ncent*5][32]
    const float *s1, *h1; // [32] scale / shift of the first attention BatchNorm
    const float *W2, *b2; // [128][32], [128]
    const float *scp, *shp, *sca, *sha;   // [128]
    long long ncent;
    float *agg;
    int lda;
    unsigned char *amax;
    float *zsel;
};

__device__ void gg_buf_st_u8(unsigned char v, gg_rsrc r, unsigned lane_bytes, unsigned uniform_bytes, int aux = 0) __asm("llvm.amdgcn.raw.buffer.store.i8");
// One wave per tile, all four 32-channel column tiles (the activation of the tile's rows, the edge records and the
// row indices are paid once per tile, not once per column tile); two waves per SIMD: 256 registers a lane, and
// (amdgpu_waves_per_eu) the accumulators stay in ordinary VGPRs -- with the default bound the compiler keeps MFMA
// results in AGPRs and pays a v_accvgpr_read per value the VALU touches.  The next tile's rows and edge records are
// in flight while this one is folded.
// Measured on the way (profiles/r5_attfwd_variants.txt): the gathers as STRUCTURED buffer loads (row index as
// vindex, stride 512 -- one instruction, no address arithmetic) cost +0.24 ms: raw loads with the byte offset
// computed by a multiply-add it is.
// TRAIN = false: the evaluation forward (every BatchNorm a fixed affine map; csrc/gridgcn_atteval.hip is the general
// form): only the maximum is kept -- no arg max, no saved pre-activations.
template <bool TRAIN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gg_k_att_pairmax(GGAttFwd p)
{
    // per channel, every value twice (the two halves of a packed operand): a1 b1 | a2 b2' | w0 w1 | w2 b   (b2' = b2 + b2c a2)
    __shared__ gg_f32x4 cst[4][32][4];
    __shared__ float cb2c[128];
    __shared__ gg_f32x4 act[2][2][4];       // [scale | shift][k half][4 x float4]
    // per wave, per lane half: gx / gy / gz / byte offset of the source row of the half's 15 edges (slot 15: a dummy),
    // slot-major so that one 16-byte read delivers four slots of a component
    __shared__ gg_f32x4 geo[4][2][2][4][4];  // [wave][buffer][lane half][gx gy gz row][four slots each]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    if (threadIdx.x < 128) {
        const int ch = threadIdx.x;
        const float a1 = p.scp[ch], b1 = p.shp[ch], a2 = p.sca[ch], b2 = fmaf(p.b2[ch], p.sca[ch], p.sha[ch]);
        const float w0 = p.Wg ? p.Wg[ch] : 0.f, w1 = p.Wg ? p.Wg[128 + ch] : 0.f, w2 = p.Wg ? p.Wg[256 + ch] : 0.f;
        gg_f32x4 v;
        v.x = a1; v.y = a1; v.z = b1; v.w = b1; cst[ch >> 5][ch & 31][0] = v;
        v.x = a2; v.y = a2; v.z = b2; v.w = b2; cst[ch >> 5][ch & 31][1] = v;
        v.x = w0; v.y = w0; v.z = w1; v.w = w1; cst[ch >> 5][ch & 31][2] = v;
        v.x = w2; v.y = w2; v.z = p.b[ch]; v.w = p.b[ch]; cst[ch >> 5][ch & 31][3] = v;
        cb2c[ch] = p.b2[ch];
    } else if (threadIdx.x < 128 + 32) {
        const int k = threadIdx.x - 128;
        ((float *)act)[k] = p.s1[k];
        ((float *)act)[32 + k] = p.h1[k];
    }
    for (int i = threadIdx.x; i < 4 * 2 * 2 * 4 * 4 * 4; i += 256) ((float *)geo)[i] = 0.f;
    // B operand: W2^T, step m contracts k = m (lane half 0) and k = 16 + m (lane half 1)
    float wb[4][16];
#pragma unroll
    for (int ct = 0; ct < 4; ct++) {
        const float *wr = p.W2 + (size_t)(32 * ct + j) * 32 + h * 16;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const gg_f32x4 v = gg_ld_f4(wr + 4 * q);
            wb[ct][4 * q] = v.x; wb[ct][4 * q + 1] = v.y; wb[ct][4 * q + 2] = v.z; wb[ct][4 * q + 3] = v.w;
        }
    }
    __syncthreads();
    // (32-bit indices throughout: gg_att_fwd_ok bounds the edge count and every byte offset below 2^32)
    const unsigned ncent = (unsigned)p.ncent, ntile = (ncent + 5) / 6, E = ncent * 5;
    const int rows = p.B * p.Nsrc;
    const gg_rsrc ry = gg_make_rsrc(p.Ysrc);
    const gg_rsrc rz1 = gg_make_rsrc(p.Z1), ratt = gg_make_rsrc(p.att16), rnb = gg_make_rsrc(p.nebidx);
    const gg_rsrc ragg = gg_make_rsrc_n(p.agg, ncent * (unsigned)p.lda * 4u);
    const gg_rsrc rzp = gg_make_rsrc_n(p.zsel, ncent * 512u);
    const gg_rsrc rza = gg_make_rsrc_n(TRAIN ? p.zsel + p.ncent * 128 : p.zsel, ncent * 512u);   // (evaluation: no zsel at all)
    const gg_rsrc ram = gg_make_rsrc_n(p.amax, ncent * 128u);
    // the A row of this lane: slot s_i of lane half h_i
    const int si = (j & 3) + 4 * (j >> 3), hi = (j >> 2) & 1;
    const unsigned arow = hi * 15 + (si < 15 ? si : 14);
    // the staging lanes (0..29): edge `lane` of the tile -> half lane / 15, slot lane % 15
    const int sh_ = lane >= 15 ? 1 : 0, ss_ = lane - 15 * sh_;
    const int gso = sh_ * 64 + (ss_ & 15);
    const unsigned slane = lane < 30 ? lane : 29;
    const float NEG = -__builtin_inff();
    const unsigned jb = (unsigned)j * 4;
    typedef gg_f32x2 f2;
    struct In { gg_f32x4 z[4], g; int nb; };
    auto load = [&](unsigned t, In &in) {
        const unsigned eb = t * 30;
        unsigned ea = eb + arow;
        ea = ea < E ? ea : E - 1;
        const unsigned zo = ea * 128u + h * 64u;
#pragma unroll
        for (int q = 0; q < 4; q++) in.z[q] = gg_buf_ld4(rz1, zo + 16u * q, 0);
        unsigned e = eb + slane;
        e = e < E ? e : E - 1;
        in.g = gg_buf_ld4(ratt, e * 64u, 0);
        in.nb = (int)gg_buf_ld_u32(rnb, e * 4u, 0);
    };
    // the staging lanes' part of a tile: (gx, gy, gz, byte offset of the source row) of edge `lane` into buffer gb
    auto stage = [&](unsigned t, const In &in, float *gb) {
        // cloud of the tile's first centre (scalar), then at most one step up inside the tile
        const unsigned oc0 = t * 6, bt = oc0 / (unsigned)p.O;
        unsigned e = t * 30 + slane;
        e = e < E ? e : E - 1;
        const unsigned o = e / 5;
        const int bi = (int)bt + (o >= (bt + 1) * (unsigned)p.O ? 1 : 0);
        int flat = in.nb + bi * p.Nsrc;
        flat = flat < 0 ? 0 : (flat > rows - 1 ? rows - 1 : flat);
        if (lane < 30) {
            float *g = gb + gso;
            g[0] = in.g.y; g[16] = in.g.z; g[32] = in.g.w; g[48] = __uint_as_float((unsigned)flat * 512u);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // the sixteen slots of this lane half, one channel each (column tile ct)
    auto gather = [&](const float *gb, int ct, f2 (&y)[8]) {
        const gg_f32x4 *gq = (const gg_f32x4 *)gb + (h * 4 + 3) * 4;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const gg_f32x4 gr = gq[k];
            y[2 * k].x = gg_buf_ld(ry, __float_as_uint(gr.x) + jb + 128u * ct, 0);
            y[2 * k].y = gg_buf_ld(ry, __float_as_uint(gr.y) + jb + 128u * ct, 0);
            y[2 * k + 1].x = gg_buf_ld(ry, __float_as_uint(gr.z) + jb + 128u * ct, 0);
            y[2 * k + 1].y = gg_buf_ld(ry, __float_as_uint(gr.w) + jb + 128u * ct, 0);
        }
    };
    // Software pipeline over the wave's tiles: while tile t is folded, the rows and edge records of t + 1 are in
    // flight, its edge records are staged half-way through (into the other LDS buffer), and the gathers of ITS first
    // column tile are issued before the last MFMAs of t; inside a tile the gathers of column tile ct + 1 go out before
    // the MFMAs of ct.  Nothing the fold needs is requested less than ~1000 cycles before it is used.
    In cur;
    const unsigned tstep = gridDim.x * 4;
    // (an XCD-contiguous renumbering of the workgroups -- neighbouring tiles in one L2 -- measured neutral)
    unsigned t = blockIdx.x * 4 + wave;
    float *gcur = (float *)&geo[wave][0][0][0][0], *gnxt = (float *)&geo[wave][1][0][0][0];
    f2 yb[2][8];
    if (t < ntile) {
        load(t, cur);
        stage(t, cur, gcur);
        gather(gcur, 0, yb[0]);
    }
    for (; t < ntile; t += tstep) {
        In nxt;
        const unsigned tn = t + tstep < ntile ? t + tstep : t;
        load(tn, nxt);
        float a[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const gg_f32x4 v = gg_bnrelu4v(cur.z[q], act[0][h][q], act[1][h][q]);
            a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
        }
        // byte offsets of this lane's outputs: centre 6 t + 3 h + cs, channel j (+ 32 ct)
        const unsigned o0 = t * 6 + h * 3;
        const gg_f32x4 *gx = (const gg_f32x4 *)gcur + h * 16;
#pragma unroll
        for (int ct = 0; ct < 4; ct++) {
            if (ct < 3) gather(gcur, ct + 1, yb[(ct + 1) & 1]);
            else gather(gnxt, 0, yb[0]);
            const f2 (&y)[8] = yb[ct & 1];
            ggm_f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; r++) d[r] = 0.f;
#pragma unroll
            for (int m = 0; m < 16; m++) d = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], wb[ct][m], d, 0, 0, 0);
            const gg_f32x4 c0 = cst[ct][j][0], c1 = cst[ct][j][1], c2 = cst[ct][j][2], c3 = cst[ct][j][3];
            const f2 a1 = {c0.x, c0.y}, b1 = {c0.z, c0.w}, a2 = {c1.x, c1.y}, b2 = {c1.z, c1.w};
            const f2 w0 = {c2.x, c2.y}, w1 = {c2.z, c2.w}, w2 = {c3.x, c3.y}, wbias = {c3.z, c3.w};
            const float b2c = cb2c[32 * ct + j];
            // pairs of slots on the packed fp32 instructions
            f2 z1[TRAIN ? 8 : 1], vv[8];
#pragma unroll
            for (int k4 = 0; k4 < 4; k4++) {
                const gg_f32x4 X = gx[k4], Y = gx[4 + k4], Z = gx[8 + k4];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int k = 2 * k4 + u;
                    f2 px, py, pz, dd;
                    if (u) { px.x = X.z; px.y = X.w; py.x = Y.z; py.y = Y.w; pz.x = Z.z; pz.y = Z.w; }
                    else   { px.x = X.x; px.y = X.y; py.x = Y.x; py.y = Y.y; pz.x = Z.x; pz.y = Z.y; }
                    dd.x = d[2 * k]; dd.y = d[2 * k + 1];
                    // (the operation order of gg_k_edge_lin0_fwd: bit-identical to the statistics pass)
                    f2 zz = __builtin_elementwise_fma(px, w0, y[k]);
                    zz = __builtin_elementwise_fma(py, w1, zz);
                    zz = __builtin_elementwise_fma(pz, w2, zz);
                    zz = zz + wbias;
                    if (TRAIN) z1[TRAIN ? k : 0] = zz;
                    const f2 y1 = __builtin_elementwise_max(__builtin_elementwise_fma(zz, a1, b1), (f2)(0.f));
                    const f2 y2 = __builtin_elementwise_max(__builtin_elementwise_fma(dd, a2, b2), (f2)(0.f));
                    vv[k] = y1 * y2;
                }
            }
#pragma unroll
            for (int cs = 0; cs < 3; cs++) {
                // the FIRST neighbour that attains the maximum (what `v > best` in neighbour order selects): the
                // maximum of the five products, then the equality masks of neighbours 3 .. 0, the lowest last
                float v[5];
#pragma unroll
                for (int q = 0; q < 5; q++) {
                    const int r = cs * 5 + q;
                    v[q] = (r & 1) ? vv[r >> 1].y : vv[r >> 1].x;
                }
                const float best = fmaxf(__builtin_fmaxf(__builtin_fmaxf(v[0], v[1]), v[2]),
                                         __builtin_fmaxf(v[3], v[4]));
                const unsigned oc = o0 + cs;
                gg_buf_st(best, ragg, oc * (unsigned)(p.lda * 4) + jb + 128u * ct, 0);
                if constexpr (TRAIN) {
                    int bi = 4;
                    float zps = z1[(cs * 5 + 4) >> 1].x, zas = d[cs * 5 + 4];
                    if ((cs * 5 + 4) & 1) zps = z1[(cs * 5 + 4) >> 1].y;
#pragma unroll
                    for (int q = 3; q >= 0; q--) {
                        const int r = cs * 5 + q;
                        const bool m = v[q] == best;
                        const float zp = (r & 1) ? z1[r >> 1].y : z1[r >> 1].x;
                        bi = m ? q : bi; zps = m ? zp : zps; zas = m ? d[r] : zas;
                    }
                    gg_buf_st_u8((unsigned char)bi, ram, oc * 128u + j + 32u * ct, 0);
                    gg_buf_st(zps, rzp, oc * 512u + jb + 128u * ct, 0);
                    gg_buf_st(zas + b2c, rza, oc * 512u + jb + 128u * ct, 0);
                }
            }
            if (ct == 1) stage(tn, nxt, gnxt);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) cur.z[q] = nxt.z[q];
        float *sw = gcur; gcur = gnxt; gnxt = sw;
    }
}

// ---- S1 = sum a1, S2 = sum a1 a1^T of a1 = relu(Z1 * scale + shift), Z1 [E, 32] ------------------------------
// part [workgroups][17][64] fp64: rows 0..15 the C/D registers of S2 (lane l: column l&31, row (r&3)+8(r>>2)+4(l>>5)),
// row 16 the lane's share of S1[l&31].
#define GG_MOM_FLUSH 8           // batches of 16 row pairs between two fp64 folds (256 rows)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gg_k_att_moments(
    const float *__restrict__ Z1, const float *__restrict__ s1, const float *__restrict__ h1, unsigned E,
    double *__restrict__ part)
{
    __shared__ double sl[4][17][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    const float sc = s1[lane & 31], sh = h1[lane & 31];
    const unsigned npair = (E + 1) / 2;
    unsigned per = (npair + nw - 1) / nw;
    per = (per + 15) & ~15u;
    const unsigned lo = gw * per < npair ? gw * per : npair;
    const unsigned hi = lo + per < npair ? lo + per : npair;
    // whole batches of 16 pairs whose 32 rows all exist; what is left (at most one batch) takes the predicated path
    const unsigned nfull = (hi - lo) / 16 - ((hi - lo) % 16 == 0 && hi == npair && (E & 1) && hi > lo ? 1 : 0);
    const gg_rsrc rz = gg_make_rsrc(Z1);
    ggm_f32x16 acc;
    double accd[16], s1d = 0.0;
    float s1f = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc[r] = 0.f; accd[r] = 0.0; }
    // lane l reads element l of each 64-float row pair: one coalesced 256-byte load per MFMA step
    auto ld = [&](unsigned q, float (&v)[16]) {
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = gg_buf_ld(rz, q * 256u + lane * 4u + 256u * i, 0);
    };
    auto fold = [&]() {
#pragma unroll
        for (int r = 0; r < 16; r++) { accd[r] += (double)acc[r]; acc[r] = 0.f; }
        s1d += (double)s1f;
        s1f = 0.f;
    };
    float v[16];
    if (nfull) ld(lo, v);
    unsigned nb = 0;
    for (unsigned b = 0; b < nfull; b++) {
        const unsigned q = lo + 16 * b;
        float vn[16];
        if (b + 1 < nfull) ld(q + 16, vn);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            // (two alternating accumulators: no faster -- the chain of dependent MFMAs is not what bounds this)
            const float a = fmaxf(fmaf(v[i], sc, sh), 0.f);
            s1f += a;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc, 0, 0, 0);
        }
        if (++nb == GG_MOM_FLUSH) { nb = 0; fold(); }
        if (b + 1 < nfull) {
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = vn[i];
        }
    }
    for (unsigned q = lo + 16 * nfull; q < hi; q++) {         // the ragged end: pair by pair
        const unsigned row = 2 * q + (lane >> 5);
        const float z = gg_buf_ld(rz, (row < E ? row : E - 1) * 128u + (lane & 31) * 4u, 0);
        const float a = row < E ? fmaxf(fmaf(z, sc, sh), 0.f) : 0.f;
        s1f += a;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc, 0, 0, 0);
    }
    fold();
#pragma unroll
    for (int r = 0; r < 16; r++) sl[wave][r][lane] = accd[r];
    sl[wave][16][lane] = s1d;
    __syncthreads();
    for (int i = threadIdx.x; i < 17 * 64; i += 256) {
        const double *s = &sl[0][0][0] + i;
        part[(size_t)blockIdx.x * (17 * 64) + i] = ((s[0] + s[17 * 64]) + s[2 * 17 * 64]) + s[3 * 17 * 64];
    }
}

// mom[17][64] = sum over the workgroups, fixed order (block = row, 16 slices of them, then the slices in order)
__global__ __launch_bounds__(1024) void gg_k_att_moments_reduce(const double *__restrict__ part, int nw,
                                                                double *__restrict__ mom)
{
    __shared__ double sl[16][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6, row = blockIdx.x;
    const int per = (nw + 15) / 16;
    const int w0 = slice * per, w1 = w0 + per < nw ? w0 + per : nw;
    double s = 0.0;
    for (int w = w0; w < w1; w++) s += part[((size_t)w * 17 + row) * 64 + lane];
    sl[slice][lane] = s;
    __syncthreads();
    if (threadIdx.x < 64) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 16; i++) t += sl[i][lane];
        mom[row * 64 + lane] = t;
    }
}

// the second attention BatchNorm from the moments: one wave per output channel (lane l: column l&31 of S2, sixteen
// of its rows), the 64 partial quadratic forms summed in a fixed butterfly order
__global__ __launch_bounds__(64) void gg_k_att_bn2_from_moments(const double *__restrict__ mom,
                                                                const float *__restrict__ W2,
                                                                const float *__restrict__ b2,
                                                                const float *__restrict__ gamma,
                                                                const float *__restrict__ beta, long long E, float eps,
                                                                float momentum, float *scale, float *shift,
                                                                float *mean, float *rstd, float *run_mean,
                                                                float *run_var, long long *nbt, double *sums)
{
    const int c = blockIdx.x, l = threadIdx.x;
    const float *w = W2 + c * 32;
    const double wj = (double)w[l & 31];
    // lane l holds S2[ggm_row(r, l)][l & 31] in mom[r][l]
    double q = 0.0;
#pragma unroll
    for (int r = 0; r < 16; r++) q += (double)w[ggm_row(r, l)] * mom[r * 64 + l];
    q *= wj;
    double m = wj * mom[1024 + l];
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        q += __shfl_xor(q, s, 64);
        m += __shfl_xor(m, s, 64);
    }
    if (l == 0) {
        const double b = (double)b2[c], n = (double)E;
        const double sz = m + n * b, szz = q + 2.0 * b * m + n * b * b;
        if (sums) { sums[c] = sz; sums[128 + c] = szz; }
        if (c == 0 && nbt) nbt[0] += 1;
        gg_bn_fin_write(sz, szz, c, gamma, beta, E, eps, momentum, scale, shift, mean, rstd, run_mean, run_var);
    }
}

static int g_att_eval_tile = 1;         // GRIDGCN_OPT_ATT_EVAL_TILE
void gg_set_att_eval_tile(int v) { g_att_eval_tile = v ? 1 : 0; }
int gg_get_att_eval_tile() { return g_att_eval_tile; }

static int gg_att_moments_grid(long long E)
{
    // >= 256 rows per wave, two workgroups per CU (four: 108 us against 93)
    long long nb = (E + 256 * 4 - 1) / (256 * 4);
    return (int)(nb < 1 ? 1 : (nb > 512 ? 512 : nb));
}

int gg_att_moments_grid_of(long long E) { return gg_att_moments_grid(E); }

bool gg_att_fwd_ok(long long ncent, int O, int P, int cin, int C, int lda, long long rows)
{
    return P == 5 && cin == 32 && C == 128 && ncent >= 7 && ncent < (1ll << 22) && O >= 6 && lda >= 128 && rows >= 1 &&
           rows < (1ll << 23) && ncent * (long long)lda * 4 < (1ll << 32);
}

size_t gg_att_moments_workspace(long long E)
{
    return ((size_t)gg_att_moments_grid(E) * 17 * 64 + 17 * 64) * sizeof(double);
}

int gg_att_bn2_moments(const float *Z1, const float *s1, const float *h1, const float *W2, const float *b2,
                       const float *gamma, const float *beta, long long E, float eps, float momentum, float *scale,
                       float *shift, float *mean, float *rstd, float *run_mean, float *run_var, long long *nbt,
                       double *sums, void *ws, hipStream_t st)
{
    const int grid = gg_att_moments_grid(E);
    double *part = (double *)ws, *mom = part + (size_t)grid * 17 * 64;
    gg_k_att_moments<<<grid, 256, 0, st>>>(Z1, s1, h1, (unsigned)E, part);
    gg_k_att_moments_reduce<<<17, 1024, 0, st>>>(part, grid, mom);
    gg_k_att_bn2_from_moments<<<128, 64, 0, st>>>(mom, W2, b2, gamma, beta, E, eps, momentum, scale, shift, mean, rstd,
                                                 run_mean, run_var, nbt, sums);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_att_pairmax(const GGAttFwd &p, hipStream_t st)
{
    const long long ntile = (p.ncent + 5) / 6;
    long long nb = (ntile + 3) / 4;
    nb = nb > 512 ? 512 : nb;                                // two workgroups (of four waves) per CU
    if (p.amax) gg_k_att_pairmax<true><<<(int)nb, 256, 0, st>>>(p);
    else gg_k_att_pairmax<false><<<(int)nb, 256, 0, st>>>(p);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_att_pairmax_args(const float *Ysrc, const int *nebidx, const float *att16, const float *Wg, const float *b,
                        int B, int Nsrc, int O, const float *Z1, const float *s1, const float *h1, const float *W2,
                        const float *b2, const float *scp, const float *shp, const float *sca, const float *sha,
                        long long ncent, float *agg, int lda, unsigned char *amax, float *zsel, hipStream_t st)
{
    GGAttFwd p;
    p.Ysrc = Ysrc; p.nebidx = nebidx; p.att16 = att16; p.Wg = Wg; p.b = b;
    p.Nsrc = Nsrc; p.O = O; p.B = B;
    p.Z1 = Z1; p.s1 = s1; p.h1 = h1; p.W2 = W2; p.b2 = b2;
    p.scp = scp; p.shp = shp; p.sca = sca; p.sha = sha;
    p.ncent = ncent; p.agg = agg; p.lda = lda; p.amax = amax; p.zsel = zsel;
    return gg_att_pairmax(p, st);
}
