// gridgcn_pairmax.hip -- attention product + max over the P neighbours, training mode (gfx950).
//
// Reference: pair = att * nf (gcn_module_g_att.py:167) and Pooling(max, kernel=(1,P)) (:57-59),
// each preceded by the BatchNorm+ReLU of the last pt / att layer (utils/ops.py:149-158): five
// tensors of [B,C,O,P] are written and re-read, and again in backward.  Here the two pre-BatchNorm
// outputs Zpt, Zatt [E,C] are read ONCE:
//   gg_k_pairmax_fwd   agg[o,c] = max_p relu(bn(Zpt))*relu(bn(Zatt)), amax[o,c] = arg max (first)
//   gg_k_pairmax_bwd   per (centre, channel): gradients w.r.t. the two post-ReLU activations at the
//                      arg-max edge only (everything else is zero) + the BatchNorm-backward sums of
//                      both last layers -- a [B*O, C] sized pass instead of four [E, C] passes.
// The dense gradient is never materialised: gg_k_linear_bwd rebuilds it from (amax, gval) while
// staging its tile.
#include <hip/hip_runtime.h>
#include "../../include/gridgcn.h"
#include "gridgcn_train.h"

__global__ __launch_bounds__(256) void gg_k_pairmax_fwd(const float *__restrict__ Zp,
                                                        const float *__restrict__ Za,
                                                        const float *__restrict__ scp,
                                                        const float *__restrict__ shp,
                                                        const float *__restrict__ sca,
                                                        const float *__restrict__ sha,
                                                        long long ncent, int P, int C,
                                                        float *__restrict__ agg,
                                                        unsigned char *__restrict__ amax,
                                                        float *__restrict__ zsel, int lda)
{
    const long long total = ncent * C;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total;
         t += (long long)gridDim.x * 256) {
        const long long o = t / C;
        const int c = (int)(t - o * C);
        const float a1 = scp[c], b1 = shp[c], a2 = sca[c], b2 = sha[c];
        const float *zp = Zp + (o * P) * C + c, *za = Za + (o * P) * C + c;
        float best = -__builtin_inff(), zps = zp[0], zas = za[0];
        int bi = 0;
        for (int p = 0; p < P; p++) {
            const float z1 = zp[(size_t)p * C], z2 = za[(size_t)p * C];
            float y1 = fmaxf(z1 * a1 + b1, 0.f);
            float y2 = fmaxf(z2 * a2 + b2, 0.f);
            float v = y1 * y2;
            if (v > best) { best = v; bi = p; zps = z1; zas = z2; }
        }
        agg[o * lda + c] = best;
        amax[t] = (unsigned char)bi;
        if (zsel) {              // the two pre-activations at the arg max: backward needs no gather
            zsel[t] = zps;
            zsel[total + t] = zas;
        }
    }
}

// Zp of a single-layer point MLP is never stored (gridgcn_edgelin.hip): it is recomputed here,
// bit-identical to the statistics pass, from the per-source-point product and geo_vec:
//   zp[e, c] = ((Ysrc[src(e), c] + gx*Wg[0,c]) + gy*Wg[1,c]) + gz*Wg[2,c] + b[c]
struct GGPtRecompute {
    const float *Ysrc;    // [B*Nsrc][C]   (nullptr: no feature term)
    const int *nebidx;    // [ncent*P]
    const float *att16;   // [ncent*P][16], geo_vec at 1..3
    const float *Wg;      // [3][C] or nullptr
    const float *b;       // [C]
    int Nsrc, O, B;
};

// PF > 0: the number of neighbours is the compile-time constant PF (the up layers: 5) and the loop
// over them is unrolled, so that the index -> source row chains of ALL neighbours are in flight
// together instead of one after the other
// ZB: Za holds bf16 values (gridgcn_linear_fwd_direct_ld zfmt 1)
__device__ __forceinline__ float4 gg_ld4_bf16(const void *p)
{
    const uint2 u = *(const uint2 *)p;
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                       __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}

template <int PF, bool ZB = false>
__global__ __launch_bounds__(256) void gg_k_pairmax_fwd4_src(GGPtRecompute r,
                                                             const float *__restrict__ Za,
                                                             const float *__restrict__ scp,
                                                             const float *__restrict__ shp,
                                                             const float *__restrict__ sca,
                                                             const float *__restrict__ sha,
                                                             long long ncent, int P, int C,
                                                             float *__restrict__ agg,
                                                             unsigned char *__restrict__ amax,
                                                             float *__restrict__ zsel, int lda)
{
    const int C4 = C >> 2;
    const long long total4 = ncent * C4;
    const long long rows = (long long)r.B * r.Nsrc;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total4;
         t += (long long)gridDim.x * 256) {
        const long long o = t / C4;
        const int c = (int)(t - o * C4) * 4;
        const int bi = (int)(o / r.O);
        const float4 a1 = *(const float4 *)(scp + c), b1 = *(const float4 *)(shp + c);
        const float4 a2 = *(const float4 *)(sca + c), b2 = *(const float4 *)(sha + c);
        const float4 bb = *(const float4 *)(r.b + c);
        float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0, w2 = w0;
        if (r.Wg) {
            w0 = *(const float4 *)(r.Wg + c);
            w1 = *(const float4 *)(r.Wg + C + c);
            w2 = *(const float4 *)(r.Wg + 2 * C + c);
        }
        const float *za = Za + (o * (PF > 0 ? PF : P)) * C + c;
        const unsigned short *zah = (const unsigned short *)Za + (o * (PF > 0 ? PF : P)) * C + c;
        auto ldza = [&](int p) -> float4 {
            if constexpr (ZB) return gg_ld4_bf16(zah + (size_t)p * C);
            else return *(const float4 *)(za + (size_t)p * C);
        };
        float best[4], zps[4], zas[4];
        int bi4[4];
        const float a1v[4] = {a1.x, a1.y, a1.z, a1.w}, b1v[4] = {b1.x, b1.y, b1.z, b1.w};
        const float a2v[4] = {a2.x, a2.y, a2.z, a2.w}, b2v[4] = {b2.x, b2.y, b2.z, b2.w};
        const float w0v[4] = {w0.x, w0.y, w0.z, w0.w}, w1v[4] = {w1.x, w1.y, w1.z, w1.w};
        const float w2v[4] = {w2.x, w2.y, w2.z, w2.w}, bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
        for (int i = 0; i < 4; i++) { best[i] = -__builtin_inff(); bi4[i] = 0; zps[i] = 0.f; zas[i] = 0.f; }
        const int PN = PF > 0 ? PF : P;
        // (no branch around the dependent row load: a conditional load made the compiler wait for
        //  every outstanding load -- s_waitcnt vmcnt(0) -- once per neighbour)
        const bool hasY = r.Ysrc != nullptr;
        const float *ysrc = hasY ? r.Ysrc + c : r.b;            // no source term: any valid address
        const long long ystride = hasY ? C : 0;
        auto fold = [&](int p, const float4 g, const float4 yy, const float4 z2) {
            const float4 y = hasY ? yy : make_float4(0.f, 0.f, 0.f, 0.f);
            const float yv[4] = {y.x, y.y, y.z, y.w}, z2v[4] = {z2.x, z2.y, z2.z, z2.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float z1 = yv[i];
                z1 = fmaf(g.y, w0v[i], z1);
                z1 = fmaf(g.z, w1v[i], z1);
                z1 = fmaf(g.w, w2v[i], z1);
                z1 += bv[i];
                const float y1 = fmaxf(z1 * a1v[i] + b1v[i], 0.f);
                const float y2 = fmaxf(z2v[i] * a2v[i] + b2v[i], 0.f);
                const float v = y1 * y2;
                const bool upd = v > best[i];
                if (upd || p == 0) { zps[i] = z1; zas[i] = z2v[i]; }
                if (upd) { best[i] = v; bi4[i] = p; }
            }
        };
        if constexpr (PF > 0) {
            // all PF indices, then every independent load, then the PF dependent rows together
            int nb[PF];
            float4 gq[PF], zq[PF], yq[PF];
#pragma unroll
            for (int p = 0; p < PF; p++) nb[p] = r.nebidx[o * PF + p];
#pragma unroll
            for (int p = 0; p < PF; p++) {
                gq[p] = *(const float4 *)(r.att16 + (o * PF + p) * 16);       // (dist, gx, gy, gz)
                zq[p] = ldza(p);
            }
#pragma unroll
            for (int p = 0; p < PF; p++) {
                long long flat = (long long)nb[p] + (long long)bi * r.Nsrc;
                flat = flat < 0 ? 0 : (flat > rows - 1 ? rows - 1 : flat);
                yq[p] = *(const float4 *)(ysrc + flat * ystride);
            }
#pragma unroll
            for (int p = 0; p < PF; p++) fold(p, gq[p], yq[p], zq[p]);
        } else {
            for (int p = 0; p < PN; p++) {
                const long long e = o * PN + p;
                long long flat = (long long)r.nebidx[e] + (long long)bi * r.Nsrc;
                flat = flat < 0 ? 0 : (flat > rows - 1 ? rows - 1 : flat);
                const float4 g = *(const float4 *)(r.att16 + e * 16);       // (dist, gx, gy, gz)
                const float4 y = *(const float4 *)(ysrc + flat * ystride);
                const float4 z2 = ldza(p);
                fold(p, g, y, z2);
            }
        }
        const long long e = o * C + c;
        *(float4 *)(agg + o * lda + c) = make_float4(best[0], best[1], best[2], best[3]);
        *(unsigned *)(amax + e) = (unsigned)bi4[0] | ((unsigned)bi4[1] << 8) | ((unsigned)bi4[2] << 16) | ((unsigned)bi4[3] << 24);
        if (zsel) {
            *(float4 *)(zsel + e) = make_float4(zps[0], zps[1], zps[2], zps[3]);
            *(float4 *)(zsel + ncent * C + e) = make_float4(zas[0], zas[1], zas[2], zas[3]);
        }
    }
}

// same, four channels per thread with 16-byte loads (C % 4 == 0)
__global__ __launch_bounds__(256) void gg_k_pairmax_fwd4(const float *__restrict__ Zp,
                                                         const float *__restrict__ Za,
                                                         const float *__restrict__ scp,
                                                         const float *__restrict__ shp,
                                                         const float *__restrict__ sca,
                                                         const float *__restrict__ sha,
                                                         long long ncent, int P, int C,
                                                         float *__restrict__ agg,
                                                         unsigned char *__restrict__ amax,
                                                         float *__restrict__ zsel, int lda)
{
    const int C4 = C >> 2;
    const long long total4 = ncent * C4;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total4;
         t += (long long)gridDim.x * 256) {
        const long long o = t / C4;
        const int c = (int)(t - o * C4) * 4;
        const float4 a1 = *(const float4 *)(scp + c), b1 = *(const float4 *)(shp + c);
        const float4 a2 = *(const float4 *)(sca + c), b2 = *(const float4 *)(sha + c);
        const float *zp = Zp + (o * P) * C + c, *za = Za + (o * P) * C + c;
        float best[4], zps[4], zas[4];
        int bi[4];
        const float a1v[4] = {a1.x, a1.y, a1.z, a1.w}, b1v[4] = {b1.x, b1.y, b1.z, b1.w};
        const float a2v[4] = {a2.x, a2.y, a2.z, a2.w}, b2v[4] = {b2.x, b2.y, b2.z, b2.w};
#pragma unroll
        for (int i = 0; i < 4; i++) { best[i] = -__builtin_inff(); bi[i] = 0; zps[i] = zp[i]; zas[i] = za[i]; }
        for (int p = 0; p < P; p++) {
            const float4 z1 = *(const float4 *)(zp + (size_t)p * C);
            const float4 z2 = *(const float4 *)(za + (size_t)p * C);
            const float z1v[4] = {z1.x, z1.y, z1.z, z1.w}, z2v[4] = {z2.x, z2.y, z2.z, z2.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float y1 = fmaxf(z1v[i] * a1v[i] + b1v[i], 0.f);
                const float y2 = fmaxf(z2v[i] * a2v[i] + b2v[i], 0.f);
                const float v = y1 * y2;
                if (v > best[i]) { best[i] = v; bi[i] = p; zps[i] = z1v[i]; zas[i] = z2v[i]; }
            }
        }
        const long long e = o * C + c;
        *(float4 *)(agg + o * lda + c) = make_float4(best[0], best[1], best[2], best[3]);
        *(unsigned *)(amax + e) = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
        if (zsel) {
            *(float4 *)(zsel + e) = make_float4(zps[0], zps[1], zps[2], zps[3]);
            *(float4 *)(zsel + ncent * C + e) = make_float4(zas[0], zas[1], zas[2], zas[3]);
        }
    }
}

// same result with the P neighbours of a (centre, channel quad) dealt to PS lanes (p = sub, sub + PS, ...)
// and the partial maxima merged by the order of the sequential scan: the larger product wins, the
// lower p among equal ones.  For the layers whose (centre, quad) count does not fill the chip -- cfg4
// down0: 131 072 threads = 2 waves per SIMD walking 128 neighbours each, one dependent pair of loads
// after the other.  A wave holds 64 / PS quads; lane = sub * (64 / PS) + quad, so that the lanes of one
// sub read a contiguous piece of one row.
template <int PS>
__global__ __launch_bounds__(256) void gg_k_pairmax_fwd4_split(
    const float *__restrict__ Zp, const float *__restrict__ Za, const float *__restrict__ scp,
    const float *__restrict__ shp, const float *__restrict__ sca, const float *__restrict__ sha,
    long long ncent, int P, int C, float *__restrict__ agg, unsigned char *__restrict__ amax,
    float *__restrict__ zsel, int lda)
{
    constexpr int G = 64 / PS;
    const int C4 = C >> 2;
    const long long total4 = ncent * C4;
    const int lane = threadIdx.x & 63, sub = lane / G, ql = lane - sub * G;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    long long t = wave * G + ql;
    const bool valid = t < total4;
    if (!valid) t = total4 - 1;                  // (no early exit: the lane takes part in the merge)
    const long long o = t / C4;
    const int c = (int)(t - o * C4) * 4;
    const float4 a1 = *(const float4 *)(scp + c), b1 = *(const float4 *)(shp + c);
    const float4 a2 = *(const float4 *)(sca + c), b2 = *(const float4 *)(sha + c);
    const float *zp = Zp + (o * P) * C + c, *za = Za + (o * P) * C + c;
    const float a1v[4] = {a1.x, a1.y, a1.z, a1.w}, b1v[4] = {b1.x, b1.y, b1.z, b1.w};
    const float a2v[4] = {a2.x, a2.y, a2.z, a2.w}, b2v[4] = {b2.x, b2.y, b2.z, b2.w};
    float best[4], zps[4], zas[4];
    int bi[4];
    const int pfirst = sub < P ? sub : 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        best[i] = -__builtin_inff(); bi[i] = pfirst;
        zps[i] = zp[(size_t)pfirst * C + i]; zas[i] = za[(size_t)pfirst * C + i];
    }
    constexpr int UL = 4;                        // rows in flight per lane
    for (int p0 = sub; p0 < P; p0 += PS * UL) {
        float4 z1[UL], z2[UL];
#pragma unroll
        for (int u = 0; u < UL; u++) {
            const int p = p0 + PS * u < P ? p0 + PS * u : p0;
            z1[u] = *(const float4 *)(zp + (size_t)p * C);
            z2[u] = *(const float4 *)(za + (size_t)p * C);
        }
#pragma unroll
        for (int u = 0; u < UL; u++) {
            const int p = p0 + PS * u;
            if (p >= P) break;
            const float z1v[4] = {z1[u].x, z1[u].y, z1[u].z, z1[u].w}, z2v[4] = {z2[u].x, z2[u].y, z2[u].z, z2[u].w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float y1 = fmaxf(z1v[i] * a1v[i] + b1v[i], 0.f);
                const float y2 = fmaxf(z2v[i] * a2v[i] + b2v[i], 0.f);
                const float v = y1 * y2;
                if (v > best[i]) { best[i] = v; bi[i] = p; zps[i] = z1v[i]; zas[i] = z2v[i]; }
            }
        }
    }
#pragma unroll
    for (int off = G; off < 64; off <<= 1) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float ov = __shfl_xor(best[i], off, 64);
            const int op = __shfl_xor(bi[i], off, 64);
            const float oz1 = __shfl_xor(zps[i], off, 64), oz2 = __shfl_xor(zas[i], off, 64);
            const bool take = ov > best[i] || (ov == best[i] && op < bi[i]);
            best[i] = take ? ov : best[i]; bi[i] = take ? op : bi[i];
            zps[i] = take ? oz1 : zps[i]; zas[i] = take ? oz2 : zas[i];
        }
    }
    if (!valid || sub != 0) return;
    const long long e = o * C + c;
    *(float4 *)(agg + o * lda + c) = make_float4(best[0], best[1], best[2], best[3]);
    *(unsigned *)(amax + e) = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
    if (zsel) {
        *(float4 *)(zsel + e) = make_float4(zps[0], zps[1], zps[2], zps[3]);
        *(float4 *)(zsel + ncent * C + e) = make_float4(zas[0], zas[1], zas[2], zas[3]);
    }
}

// thread = (centre strip, channel); requires 256 % C == 0 (C <= 256) like gg_k_bn_bwd_reduce
__global__ __launch_bounds__(256) void gg_k_pairmax_bwd(
    const float *__restrict__ Zp, const float *__restrict__ Za, const float *__restrict__ scp,
    const float *__restrict__ shp, const float *__restrict__ mup, const float *__restrict__ rsp,
    const float *__restrict__ sca, const float *__restrict__ sha, const float *__restrict__ mua,
    const float *__restrict__ rsa, const float *__restrict__ dagg, const unsigned char *__restrict__ amax,
    long long ncent, int P, int C, float *__restrict__ gp, float *__restrict__ ga,
    double *__restrict__ sums_p, double *__restrict__ sums_a, const float *__restrict__ zsel,
    int ldd, int mask_a)
{
    __shared__ float sh[4][256];
    const int tid = threadIdx.x;
    const int rpp = 256 / C;
    const int c = tid % C, rr = tid / C;
    const float a1 = scp[c], b1 = shp[c], m1 = mup[c], r1 = rsp[c];
    const float a2 = sca[c], b2 = sha[c], m2 = mua[c], r2 = rsa[c];
    float s1p = 0.f, s2p = 0.f, s1a = 0.f, s2a = 0.f;
    for (long long o = (long long)blockIdx.x * rpp + rr; o < ncent; o += (long long)gridDim.x * rpp) {
        const long long t = o * C + c;
        float zp, za;
        if (zsel) {
            zp = zsel[t];
            za = zsel[ncent * C + t];
        } else {
            const long long e = o * P + amax[t];
            zp = Zp[e * C + c];
            za = Za[e * C + c];
        }
        const float y1 = fmaxf(zp * a1 + b1, 0.f), y2 = fmaxf(za * a2 + b2, 0.f);
        const float g = dagg[o * ldd + c];
        // gradient w.r.t. the post-ReLU activations; the ReLU mask (y > 0) is applied by the
        // consumer (gg_k_linear_bwd staging) and here for the sums
        const float g1 = g * y2, g2 = g * y1;
        const float d1 = y1 > 0.f ? g1 : 0.f, d2 = y2 > 0.f ? g2 : 0.f;
        gp[t] = g1;
        // (mask_a: the consumer has no pre-activation to mask with -- gg_att_bwd_noz -- and wants the sparse term
        //  of dZ itself: the attention layer's BatchNorm scale applied as well)
        ga[t] = mask_a ? d2 * a2 : g2;
        s1p += d1; s2p += d1 * ((zp - m1) * r1);
        s1a += d2; s2a += d2 * ((za - m2) * r2);
    }
    sh[0][tid] = s1p; sh[1][tid] = s2p; sh[2][tid] = s1a; sh[3][tid] = s2a;
    __syncthreads();
    if (tid < C) {
        for (int j = 1; j < rpp; j++) {
            s1p += sh[0][tid + j * C]; s2p += sh[1][tid + j * C];
            s1a += sh[2][tid + j * C]; s2a += sh[3][tid + j * C];
        }
        atomicAdd(&sums_p[c], (double)s1p); atomicAdd(&sums_p[C + c], (double)s2p);
        atomicAdd(&sums_a[c], (double)s1a); atomicAdd(&sums_a[C + c], (double)s2a);
    }
}

// ------------------------------------------------------------------------------------------
// BatchNorm bookkeeping on [C]-sized vectors in ONE launch each (instead of ~10 tiny element-wise
// launches per layer): batch statistics -> (scale, shift, mean, rstd) + running estimates
// (torch.nn.BatchNorm1d semantics: biased variance to normalise, unbiased for running_var).
__global__ void gg_k_bn_finalize(const double *__restrict__ sums, const float *__restrict__ gamma,
                                 const float *__restrict__ beta, long long E, float eps,
                                 float momentum, int C, float *__restrict__ scale,
                                 float *__restrict__ shift, float *__restrict__ mean,
                                 float *__restrict__ rstd, float *__restrict__ run_mean,
                                 float *__restrict__ run_var, long long *__restrict__ nbt, int tail)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt) nbt[0] += 1;                  // BatchNorm1d.num_batches_tracked
    if (c >= C) {
        // columns beyond the layer in a wider table (train_ops.RawLink): the identity
        if (c < C + tail) { scale[c] = 1.f; shift[c] = 0.f; mean[c] = 0.f; rstd[c] = 0.f; }
        return;
    }
    gg_bn_fin_write(sums[c], sums[C + c], c, gamma, beta, E, eps, momentum, scale, shift, mean, rstd, run_mean,
                    run_var);
}

// m1 = s1/E, m2 = s2/E, dbeta = s1, dgamma = s2
__global__ void gg_k_bn_bwd_finalize(const double *__restrict__ sums, long long E, int C,
                                     float *__restrict__ m1, float *__restrict__ m2,
                                     float *__restrict__ dgamma, float *__restrict__ dbeta)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double s1 = sums[c], s2 = sums[C + c];
    m1[c] = (float)(s1 / (double)E);
    m2[c] = (float)(s2 / (double)E);
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
}

// All operand layouts of one linear layer from the framework's W [C][cin] in ONE launch:
//   Wp [groups][K][32][nt]   forward B operand (K = round4(cin), ldw = C rounded to 32/64/128/256)
//   Bp [ldw]                 zero padded bias
//   Wb [ceil(cin/32)][C4][32]            tile-major W (dW / monolithic backward), C4 = round4(C)
//   Wg column blocks of 4/2/1 tiles, each [C4][32][nt]   (gg_k_linear_dx)
//   Wq  forward B operand of gg_k_linear_fwd_direct, Wdx the dX B operand of gg_k_linear_dx_direct
// The kernels see a layer of `cin` input channels; the framework's matrix has cin_w <= cin columns,
// of which the first `rot` are moved behind the others (column k of the kernels' matrix = framework
// column k + rot for k < cin_w - rot, k - (cin_w - rot) for k < cin_w, zero beyond): this is the
// "features first, geo_vec last, zero padded" row layout written by gg_k_edge_inputs_rows.
struct GGPackW {
    const float *W;
    int cin_w, rot;
    __device__ float at(int row, int k) const
    {
        if (k >= cin_w) return 0.f;
        const int src = k < cin_w - rot ? k + rot : k - (cin_w - rot);
        return W[(size_t)row * cin_w + src];
    }
};

__device__ __forceinline__ void gg_pack_linear_body(int t, const float *__restrict__ W_,
                                                    const float *__restrict__ b, int C, int cin_w,
                                                    int rot, int cin, int K, int ldw, int ndx,
                                                    float *__restrict__ Wp, float *__restrict__ Bp,
                                                    float *__restrict__ Wb, float *__restrict__ Wg,
                                                    float *__restrict__ Wq, float *__restrict__ Wdx)
{
    const int K8 = (cin + 7) & ~7;
    const GGPackW W{W_, cin_w, rot};
    if (Wdx && ndx > 0) {
        // [step s over channels][lane][NTV], channel(s, lane) as k(s, lane) below, column
        // n = t*32 + (lane&31) < ndx
        const int C8 = (C + 7) & ~7, nt = (ndx + 31) / 32;
        const int NTV = nt <= 1 ? 1 : (nt <= 2 ? 2 : (nt <= 4 ? 4 : 8));
        if (t < (C8 / 2) * 64 * NTV) {
            const int tt = t % NTV, lane = (t / NTV) % 64, s = t / (NTV * 64);
            const int c = s >> 4, ws = s & 15;
            const int kc = (C8 - 32 * c) < 32 ? (C8 - 32 * c) : 32, nq = kc >> 3;
            const int ch = 32 * c + (lane >> 5) * 4 * nq + 4 * (ws >> 2) + (ws & 3);
            const int n = tt * 32 + (lane & 31);
            Wdx[t] = (ch < C && n < ndx && n < cin) ? W.at(ch, n) : 0.f;
        }
    }
    if (Wq && t < K8 * ldw) {
        // forward B operand of the register-direct kernel (gridgcn_direct.hip):
        // [step s][lane][NT], k(s, lane) = 32c + (lane>>5)*4*nq + 4q + i
        const int NT = ldw / 32;
        const int tt = t % NT, lane = (t / NT) % 64, s = t / (NT * 64);
        const int c = s >> 4, ws = s & 15;
        const int kc = (K8 - 32 * c) < 32 ? (K8 - 32 * c) : 32, nq = kc >> 3;
        const int q = ws >> 2, i = ws & 3;
        const int k = 32 * c + (lane >> 5) * 4 * nq + 4 * q + i;
        const int col = tt * 32 + (lane & 31);
        Wq[t] = (k < cin && col < C) ? W.at(col, k) : 0.f;
    }
    if (Wp && t < K * ldw) {
        const int gw = ldw < 128 ? ldw : 128, nt = gw / 32;
        const int tt = t % nt, j = (t / nt) % 32, k = (t / (nt * 32)) % K, g = t / (nt * 32 * K);
        const int col = g * gw + tt * 32 + j;
        Wp[t] = (k < cin && col < C) ? W.at(col, k) : 0.f;
    }
    if (Bp && t < ldw) Bp[t] = t < C ? b[t] : 0.f;
    const int C4 = (C + 3) & ~3, ntile = (cin + 31) / 32;
    if (t < ntile * C4 * 32) {
        if (Wb) {
            const int j = t % 32, kc = (t / 32) % C4, tile = t / (32 * C4);
            const int col = tile * 32 + j;
            Wb[t] = (kc < C && col < cin) ? W.at(kc, col) : 0.f;
        }
        if (Wg) {
            int idx = t, done = 0, nb = 1;
            while (true) {
                const int rem = ntile - done;
                nb = rem >= 4 ? 4 : (rem >= 2 ? 2 : 1);
                const int size = C4 * 32 * nb;
                if (idx < size) break;
                idx -= size;
                done += nb;
            }
            const int tt = idx % nb, j = (idx / nb) % 32, kc = idx / (nb * 32);
            const int col = (done + tt) * 32 + j;
            Wg[t] = (kc < C && col < cin) ? W.at(kc, col) : 0.f;
        }
    }
}

__global__ void gg_k_pack_linear(const float *__restrict__ W_, const float *__restrict__ b, int C,
                                 int cin_w, int rot, int cin, int K, int ldw, int ndx,
                                 float *__restrict__ Wp, float *__restrict__ Bp,
                                 float *__restrict__ Wb, float *__restrict__ Wg,
                                 float *__restrict__ Wq, float *__restrict__ Wdx)
{
    gg_pack_linear_body(blockIdx.x * blockDim.x + threadIdx.x, W_, b, C, cin_w, rot, cin, K, ldw, ndx,
                        Wp, Bp, Wb, Wg, Wq, Wdx);
}

// every layer of a network in ONE launch: blockIdx.y = layer, its descriptor read from device memory
// (the weights change once per optimizer step; 27 pack launches per training step were ~110 us of
// 4-us kernels and their boundaries)
__global__ void gg_k_pack_linear_batch(const gridgcn_pack_desc *__restrict__ d)
{
    const gridgcn_pack_desc e = d[blockIdx.y];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= e.n) return;
    if (e.wgb) {
        // geo_vec weights + bias of a first point conv (see gridgcn_pack_desc): nothing else is packed
        if (t < 4 * e.C) {
            const int r = t / e.C, c = t - r * e.C;
            e.wgb[t] = r < 3 ? (e.geo ? e.W[(size_t)c * e.cin_w + r] : 0.f) : e.b[c];
        }
        return;
    }
    gg_pack_linear_body(t, e.W, e.b, e.C, e.cin_w, e.rot, e.cin, e.K, e.ldw, e.ndx, e.Wp, e.Bp, e.Wb,
                        e.Wg, e.Wq, e.Wdx);
}

// threads one layer's pack needs (the largest of its layouts)
static int gg_pack_threads(int C, int cin, bool wdx)
{
    const int K8 = (cin + 7) & ~7;
    const int ldw = C <= 32 ? 32 : (C <= 64 ? 64 : (C <= 128 ? 128 : 256));
    const int C4 = (C + 3) & ~3, ntile = (cin + 31) / 32;
    int n = K8 * ldw;
    if (ntile * C4 * 32 > n) n = ntile * C4 * 32;
    const int C8 = (C + 7) & ~7;
    if (C8 * 32 * 8 > n && wdx) n = C8 * 32 * 8;
    return n;
}

int gg_pack_desc_fill(gridgcn_pack_desc *e)
{
    if (!e || !e->W || (e->Bp && !e->b) || e->C < 1 || e->C > 256 || e->cin_w < 1 || e->cin < e->cin_w ||
        e->rot < 0 || e->rot > e->cin_w || e->ndx < 0 || e->ndx > 256)
        return 1;
    e->K = (e->cin + 3) & ~3;
    e->ldw = e->C <= 32 ? 32 : (e->C <= 64 ? 64 : (e->C <= 128 ? 128 : 256));
    e->n = gg_pack_threads(e->C, e->cin, e->Wdx != nullptr);
    if (e->wgb) {
        if (!e->b || (e->geo && e->cin_w < 3)) return 1;
        e->n = 4 * e->C;
    }
    return 0;
}

int gg_pack_linear_batch(const gridgcn_pack_desc *dev, int nlayers, int max_n, hipStream_t st)
{
    if (!dev || nlayers < 1 || max_n < 1) return 1;
    gg_k_pack_linear_batch<<<dim3((max_n + 255) / 256, nlayers), 256, 0, st>>>(dev);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_pack_linear(const float *W, const float *b, int C, int cin_w, int rot, int cin, int ndx,
                   float *Wp, float *Bp, float *Wb, float *Wg, float *Wq, float *Wdx,
                   hipStream_t st)
{
    if (C < 1 || C > 256 || cin_w < 1 || cin < cin_w || rot < 0 || rot > cin_w || ndx < 0 ||
        ndx > 256)
        return 1;
    const int K = (cin + 3) & ~3, K8 = (cin + 7) & ~7;
    const int ldw = C <= 32 ? 32 : (C <= 64 ? 64 : (C <= 128 ? 128 : 256));
    const int C4 = (C + 3) & ~3, ntile = (cin + 31) / 32;
    int n = K8 * ldw;
    if (ntile * C4 * 32 > n) n = ntile * C4 * 32;
    const int C8 = (C + 7) & ~7;
    if (C8 * 32 * 8 > n && Wdx) n = C8 * 32 * 8;
    gg_k_pack_linear<<<(n + 255) / 256, 256, 0, st>>>(W, b, C, cin_w, rot, cin, K, ldw, ndx, Wp, Bp,
                                                      Wb, Wg, Wq, Wdx);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_bn_finalize(const double *sums, const float *gamma, const float *beta, long long E,
                   float eps, float momentum, int C, float *scale, float *shift, float *mean,
                   float *rstd, float *run_mean, float *run_var, long long *nbt, hipStream_t st, int tail)
{
    gg_k_bn_finalize<<<(C + tail + 255) / 256, 256, 0, st>>>(sums, gamma, beta, E, eps, momentum, C, scale,
                                                             shift, mean, rstd, run_mean, run_var, nbt, tail);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_bn_bwd_finalize(const double *sums, long long E, int C, float *m1, float *m2, float *dgamma,
                       float *dbeta, hipStream_t st)
{
    gg_k_bn_bwd_finalize<<<(C + 255) / 256, 256, 0, st>>>(sums, E, C, m1, m2, dgamma, dbeta);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_pairmax_fwd_src(const float *Ysrc, const int *nebidx, const float *att16, const float *Wg,
                       const float *b, int B, int Nsrc, int O, const float *Za, const float *scp,
                       const float *shp, const float *sca, const float *sha, long long ncent, int P,
                       int C, float *agg, int lda, unsigned char *amax, float *zsel, int za_bf16,
                       hipStream_t st)
{
    if ((C & 3) || (lda & 3)) return 1;
    GGPtRecompute r;
    r.Ysrc = Ysrc; r.nebidx = nebidx; r.att16 = att16; r.Wg = Wg; r.b = b;
    r.Nsrc = Nsrc; r.O = O; r.B = B;
    long long nb = (ncent * (C / 4) + 255) / 256;
    int grid = (int)(nb < 1 ? 1 : (nb > 262144 ? 262144 : nb));
    if (P == 5 && za_bf16)
        gg_k_pairmax_fwd4_src<5, true><<<grid, 256, 0, st>>>(r, Za, scp, shp, sca, sha, ncent, P, C, agg,
                                                             amax, zsel, lda);
    else if (za_bf16)
        gg_k_pairmax_fwd4_src<0, true><<<grid, 256, 0, st>>>(r, Za, scp, shp, sca, sha, ncent, P, C, agg,
                                                             amax, zsel, lda);
    else if (P == 5)
        gg_k_pairmax_fwd4_src<5><<<grid, 256, 0, st>>>(r, Za, scp, shp, sca, sha, ncent, P, C, agg,
                                                       amax, zsel, lda);
    else
        gg_k_pairmax_fwd4_src<0><<<grid, 256, 0, st>>>(r, Za, scp, shp, sca, sha, ncent, P, C, agg,
                                                       amax, zsel, lda);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_pairmax_split = 0;          // GRIDGCN_OPT_PAIRMAX_SPLIT: 0 = by size

int gg_pairmax_fwd(const float *Zp, const float *Za, const float *scp, const float *shp,
                   const float *sca, const float *sha, long long ncent, int P, int C, float *agg,
                   int lda, unsigned char *amax, float *zsel, hipStream_t st)
{
    if ((C & 3) == 0 && (lda & 3) == 0) {
        // fewer (centre, quad) threads than 8 waves per SIMD: the neighbours of one are dealt to 2-8 lanes
        const long long total4 = ncent * (C / 4);
        int ps = 1;
        while (ps < 8 && total4 * ps < 524288 && P >= 8 * ps) ps *= 2;
        if (gg_pairmax_split) ps = gg_pairmax_split;
        if (ps > 1) {
            const long long nwave = (total4 + 64 / ps - 1) / (64 / ps);
            const long long nbs = (nwave + 3) / 4;
            if (nbs <= 0x7fffffffll) {
                if (ps == 2) gg_k_pairmax_fwd4_split<2><<<(int)nbs, 256, 0, st>>>(Zp, Za, scp, shp, sca, sha, ncent, P, C, agg, amax, zsel, lda);
                else if (ps == 4) gg_k_pairmax_fwd4_split<4><<<(int)nbs, 256, 0, st>>>(Zp, Za, scp, shp, sca, sha, ncent, P, C, agg, amax, zsel, lda);
                else gg_k_pairmax_fwd4_split<8><<<(int)nbs, 256, 0, st>>>(Zp, Za, scp, shp, sca, sha, ncent, P, C, agg, amax, zsel, lda);
                return hipGetLastError() == hipSuccess ? 0 : 3;
            }
        }
        long long nb = (ncent * (C / 4) + 255) / 256;
        int grid = (int)(nb < 1 ? 1 : (nb > 262144 ? 262144 : nb));
        gg_k_pairmax_fwd4<<<grid, 256, 0, st>>>(Zp, Za, scp, shp, sca, sha, ncent, P, C, agg, amax,
                                                zsel, lda);
        return hipGetLastError() == hipSuccess ? 0 : 3;
    }
    long long nb = (ncent * C + 255) / 256;
    int grid = (int)(nb < 1 ? 1 : (nb > 262144 ? 262144 : nb));
    gg_k_pairmax_fwd<<<grid, 256, 0, st>>>(Zp, Za, scp, shp, sca, sha, ncent, P, C, agg, amax,
                                           zsel, lda);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_pairmax_bwd(const float *Zp, const float *Za, const float *scp, const float *shp,
                   const float *mup, const float *rsp, const float *sca, const float *sha,
                   const float *mua, const float *rsa, const float *dagg, const unsigned char *amax,
                   long long ncent, int P, int C, int ldd, float *gp, float *ga, double *sums_p,
                   double *sums_a, const float *zsel, int mask_a, hipStream_t st)
{
    if (C > 256 || 256 % C != 0) return 1;
    const int rpp = 256 / C;
    long long nb = (ncent + rpp * 8 - 1) / (rpp * 8);
    int grid = (int)(nb < 1 ? 1 : (nb > 4096 ? 4096 : nb));
    gg_k_pairmax_bwd<<<grid, 256, 0, st>>>(Zp, Za, scp, shp, mup, rsp, sca, sha, mua, rsa, dagg,
                                           amax, ncent, P, C, gp, ga, sums_p, sums_a, zsel, ldd, mask_a);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
