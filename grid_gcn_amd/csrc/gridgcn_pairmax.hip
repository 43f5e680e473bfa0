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

__global__ __launch_bounds__(256) void gg_k_pairmax_fwd(const float *__restrict__ Zp,
                                                        const float *__restrict__ Za,
                                                        const float *__restrict__ scp,
                                                        const float *__restrict__ shp,
                                                        const float *__restrict__ sca,
                                                        const float *__restrict__ sha,
                                                        long long ncent, int P, int C,
                                                        float *__restrict__ agg,
                                                        int *__restrict__ amax)
{
    const long long total = ncent * C;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total;
         t += (long long)gridDim.x * 256) {
        const long long o = t / C;
        const int c = (int)(t - o * C);
        const float a1 = scp[c], b1 = shp[c], a2 = sca[c], b2 = sha[c];
        const float *zp = Zp + (o * P) * C + c, *za = Za + (o * P) * C + c;
        float best = -__builtin_inff();
        int bi = 0;
        for (int p = 0; p < P; p++) {
            float y1 = fmaxf(zp[(size_t)p * C] * a1 + b1, 0.f);
            float y2 = fmaxf(za[(size_t)p * C] * a2 + b2, 0.f);
            float v = y1 * y2;
            if (v > best) { best = v; bi = p; }
        }
        agg[t] = best;
        amax[t] = bi;
    }
}

// thread = (centre strip, channel); requires 256 % C == 0 (C <= 256) like gg_k_bn_bwd_reduce
__global__ __launch_bounds__(256) void gg_k_pairmax_bwd(
    const float *__restrict__ Zp, const float *__restrict__ Za, const float *__restrict__ scp,
    const float *__restrict__ shp, const float *__restrict__ mup, const float *__restrict__ rsp,
    const float *__restrict__ sca, const float *__restrict__ sha, const float *__restrict__ mua,
    const float *__restrict__ rsa, const float *__restrict__ dagg, const int *__restrict__ amax,
    long long ncent, int P, int C, float *__restrict__ gp, float *__restrict__ ga,
    double *__restrict__ sums_p, double *__restrict__ sums_a)
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
        const long long e = o * P + amax[t];
        const float zp = Zp[e * C + c], za = Za[e * C + c];
        const float y1 = fmaxf(zp * a1 + b1, 0.f), y2 = fmaxf(za * a2 + b2, 0.f);
        const float g = dagg[t];
        // gradient w.r.t. the post-ReLU activations; the ReLU mask (y > 0) is applied by the
        // consumer (gg_k_linear_bwd staging) and here for the sums
        const float g1 = g * y2, g2 = g * y1;
        gp[t] = g1;
        ga[t] = g2;
        const float d1 = y1 > 0.f ? g1 : 0.f, d2 = y2 > 0.f ? g2 : 0.f;
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

int gg_pairmax_fwd(const float *Zp, const float *Za, const float *scp, const float *shp,
                   const float *sca, const float *sha, long long ncent, int P, int C, float *agg,
                   int *amax, hipStream_t st)
{
    long long nb = (ncent * C + 255) / 256;
    int grid = (int)(nb < 1 ? 1 : (nb > 262144 ? 262144 : nb));
    gg_k_pairmax_fwd<<<grid, 256, 0, st>>>(Zp, Za, scp, shp, sca, sha, ncent, P, C, agg, amax);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_pairmax_bwd(const float *Zp, const float *Za, const float *scp, const float *shp,
                   const float *mup, const float *rsp, const float *sca, const float *sha,
                   const float *mua, const float *rsa, const float *dagg, const int *amax,
                   long long ncent, int P, int C, float *gp, float *ga, double *sums_p,
                   double *sums_a, hipStream_t st)
{
    if (C > 256 || 256 % C != 0) return 1;
    const int rpp = 256 / C;
    long long nb = (ncent + rpp * 8 - 1) / (rpp * 8);
    int grid = (int)(nb < 1 ? 1 : (nb > 4096 ? 4096 : nb));
    gg_k_pairmax_bwd<<<grid, 256, 0, st>>>(Zp, Za, scp, shp, mup, rsp, sca, sha, mua, rsa, dagg,
                                           amax, ncent, P, C, gp, ga, sums_p, sums_a);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
