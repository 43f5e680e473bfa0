// gridgcn_train.h -- parameter blocks / host entries of the training kernels.
#pragma once
#include <hip/hip_runtime.h>

// arg-max neighbour of a (centre, channel) pair: one byte (P <= 128 neighbours per centre)
typedef unsigned char gg_amax_t;
// four consecutive arg-max entries with ONE 32-bit load (address a multiple of 4)
__device__ __forceinline__ int4 gg_amax4(const gg_amax_t *p)
{
    const unsigned v = *(const unsigned *)p;
    return make_int4((int)(v & 255u), (int)((v >> 8) & 255u), (int)((v >> 16) & 255u), (int)(v >> 24));
}

struct GGLinFwd {
    const float *X;       // [E][cin] row-major
    const float *W;       // packed [groups][K][32][NT] (see gridgcn.h)
    const float *b;       // [>= cout]
    const float *scale;   // prologue: x <- relu(x*scale[c] + shift[c]); nullptr = identity
    const float *shift;
    float *Z;             // [E][cout]
    double *sums;         // [2][cout]: sum z, sum z^2 (accumulated; zeroed by the caller)
    long long E;
    int cin, K, ldw, cout, lda;
    // register-direct kernel only: second row source for columns [K1, K) (nullptr: X holds all K),
    // and a bias per group of P consecutive rows [E/P][cout] replacing b (nullptr: b)
    const float *X2 = nullptr;
    int K1 = 0, lda2 = 0;
    const float *rowbias = nullptr;
    int P = 0;
    int ldz = 0;          // row stride of Z (0: cout) -- Z may be the left columns of a wider buffer
    int zfmt = 0;         // 0: Z is fp32; 1: Z is bf16 (round to nearest even; ldz in elements) -- the
                          // register-direct kernel only; the statistics are those of the fp32 values
    // register-direct kernel only, optional (fin_ticket != nullptr): the layer's BatchNorm finalisation
    // (gg_bn_fin_write: scale / shift / mean / rstd, running statistics) by the last workgroup to arrive
    // instead of a launch of its own.  fin_ticket: one zero int per call.
    const float *gamma = nullptr, *beta = nullptr;
    float *fscale = nullptr, *fshift = nullptr, *fmean = nullptr, *frstd = nullptr;
    float *run_mean = nullptr, *run_var = nullptr;
    long long *nbt = nullptr;
    int *fin_ticket = nullptr;
    float eps = 0.f, momentum = 0.f;
    int fin_tail = 0;
    // register-direct kernel, one column tile, fp32 only: Dropout of the activated input rows applied while
    // they are loaded (element idx = row * K + k, gg_drop_keep; drop_thr 0 = off) -- the class-score conv of
    // the segmentation head reads fc1's raw output and no activated / dropped copy of it exists
    const unsigned long long *drop_dev = nullptr;
    unsigned drop_thr = 0, drop_lo = 0, drop_hi = 0;
    float drop_scale = 1.f;
};

struct GGLinBwd {
    const float *dY;      // [E][C] gradient w.r.t. relu(bn(Z))
    const float *Z;       // [E][C] pre-BatchNorm output of this layer
    const float *scale, *shift, *mean, *rstd;   // [C] this layer (scale = gamma*rstd)
    const float *m1, *m2;                       // [C] sum(dyr)/E, sum(dyr*zhat)/E
    const float *Aprev;   // [E][cin] raw input of this layer: Z of the previous layer, or X
    const float *pscale, *pshift, *pmean, *prstd;  // [cin] previous layer's BN (nullptr: Aprev = X)
    const float *Wb;      // W (torch layout [C][cin]) packed tile-major [ceil(cin/32)][C4][32]
    const float *Wg;      // same W packed in column blocks of 4/2/1 tiles: [block][C4][32][nt]
                          // (nullptr: no split mode; dX comes from the monolithic kernel)
    const float *Wdx;     // register-direct dX operand (gridgcn_pack_linear), nullptr: LDS-staged dX
    int ndx;              // number of leading dX columns that are needed (<= cin, <= 256)
    int ldy;              // row stride of the dense dY (>= C; the register-direct kernels only)
    int cin_w, rot;       // dW is written as [C][cin_w] in the framework's column order: kernel
                          // column k -> k + rot (k < cin_w - rot), k - (cin_w - rot) (k < cin_w)
    float *dX;            // [E][cin] gradient w.r.t. act(Aprev) (nullptr: not needed)
    float *dWpart;        // workspace [nwg][cinP][CP]
    float *dW;            // [C][cin]
    double *psums;        // [2][cin] BN-backward sums of the previous layer (zeroed by caller)
    long long E;
    int C, cin, ldd, lda;
    const gg_amax_t *amax;  // sparse upstream gradient (nullptr: dense dY): arg-max neighbour [E/P][C]
    const float *gval;    //   and its value [E/P][C]; row e = centre e/P, neighbour e%P
    int P, ncen_max;
    const unsigned long long *drop_dev;    // optional device scalar added to the dropout seed
    unsigned drop_thr, drop_lo, drop_hi;   // register-direct dX only: dX *= dropout mask of the
    float drop_scale;                      // [E][cin] input activation (gg_drop_keep), thr 0 = off
    int ldz = 0;          // row stride of Z (0: C); register-direct dX / dW kernels only
    int zfmt = 0;         // 0: Z is fp32; 1: Z is bf16 (gg_k_att_bwd_fused only)
    int nbn = 0;          // leading input columns that carry the previous layer's BatchNorm (0: all
                          // cin); beyond them the dX epilogue neither reads Aprev nor sums
    // BatchNorm-backward finalisation without a launch of its own: bsums = (s1, s2) [2][C] fp64 as
    // gridgcn_bn_relu_bwd_reduce / a dX epilogue left them.  The register-direct kernels form m1 = s1/E,
    // m2 = s2/E themselves (gg_bn_m1 / gg_bn_m2) and their dW reduce kernel writes dgamma = s2,
    // dbeta = s1 (and m1, m2); a legacy kernel on the path gets them from gg_k_bn_bwd_finalize first.
    const double *bsums = nullptr;
    float *fin_m1 = nullptr, *fin_m2 = nullptr, *fin_dgamma = nullptr, *fin_dbeta = nullptr;
    int dx_col0 = 0;      // register-direct dX: first output column of this launch (0 / 128) and
    int dx_wstride = 1;   //   float4 stride while staging Wdx (2: one half of an 8-tile layout)
    int rt;               // rows per workgroup tile of gg_k_linear_dw (32/64/96/128)
    unsigned t1[4];       // per wave: up to 3 GEMM1 column tiles, one byte each, 0xff = none
    unsigned t2[4][3];    // per wave: up to 12 GEMM2 (m,n) pair ids, one byte each, 0xff = none
};

#ifdef __HIPCC__
__device__ __forceinline__ float gg_bn_m1(const GGLinBwd &p, int c)
{
    return p.bsums ? (float)(p.bsums[c] / (double)p.E) : p.m1[c];
}
__device__ __forceinline__ float gg_bn_m2(const GGLinBwd &p, int c)
{
    return p.bsums ? (float)(p.bsums[p.C + c] / (double)p.E) : p.m2[c];
}
// both at once: the two loads (and, with bsums, the two divisions) issue together -- called one after the other
// they were two memory round trips in a row at the start of every dX / dW launch
__device__ __forceinline__ void gg_bn_m12(const GGLinBwd &p, int c, float &m1, float &m2)
{
    if (p.bsums) {
        const double s1 = p.bsums[c], s2 = p.bsums[p.C + c];
        m1 = (float)(s1 / (double)p.E);
        m2 = (float)(s2 / (double)p.E);
    } else {
        const float a = p.m1[c], b = p.m2[c];
        m1 = a;
        m2 = b;
    }
}
// BatchNorm bookkeeping of channel c from its two sums: what gg_k_bn_finalize writes (utils/ops.py:141-158,
// BatchNorm(eps, momentum, fix_gamma=False); running_var takes the unbiased variance as torch / MXNet do)
__device__ __forceinline__ void gg_bn_fin_write(double s1, double s2, int c, const float *gamma,
                                                const float *beta, long long E, float eps, float momentum,
                                                float *scale, float *shift, float *mean, float *rstd,
                                                float *run_mean, float *run_var)
{
    const double m = s1 / (double)E;
    double v = s2 / (double)E - m * m;
    if (v < 0.0) v = 0.0;
    const float mf = (float)m, vf = (float)v;
    const float rs = rsqrtf(vf + eps);
    const float sc = gamma[c] * rs;
    scale[c] = sc;
    shift[c] = beta[c] - mf * sc;
    mean[c] = mf;
    rstd[c] = rs;
    if (run_mean) {
        const float unb = vf * ((float)E / (float)(E > 1 ? E - 1 : 1));
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mf;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * unb;
    }
}
// tail of a dW reduce kernel (one thread per channel): the vectors gg_k_bn_bwd_finalize would write
__device__ __forceinline__ void gg_bn_bwd_fin_write(const double *bsums, long long E, int C, int c, float *m1,
                                                    float *m2, float *dgamma, float *dbeta)
{
    const double s1 = bsums[c], s2 = bsums[C + c];
    m1[c] = (float)(s1 / (double)E);
    m2[c] = (float)(s2 / (double)E);
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
}
#endif

int gg_bn_bwd_finalize(const double *sums, long long E, int C, float *m1, float *m2, float *dgamma,
                       float *dbeta, hipStream_t st);       // gridgcn_pairmax.hip
int gg_linear_fwd(const GGLinFwd &p, hipStream_t st);
int gg_linear_fwd_direct(const GGLinFwd &p, hipStream_t st);   // gridgcn_direct.hip
int gg_linear_dx_direct(const GGLinBwd &p, hipStream_t st);    // 1 = shape not supported
int gg_linear_dw_direct(const GGLinBwd &p, hipStream_t st);
size_t gg_linear_dw_direct_workspace(long long E, int cin, int C);   // 0 = shape not supported
// gridgcn_attfwd.hip: the forward of the attention pair product / max without the second conv's pre-activation
bool gg_att_fwd_ok(long long ncent, int O, int P, int cin, int C, int lda, long long rows);
size_t gg_att_moments_workspace(long long E);
void gg_set_att_eval_tile(int v);
int gg_get_att_eval_tile();
int gg_att_bn2_moments(const float *Z1, const float *s1, const float *h1, const float *W2, const float *b2,
                       const float *gamma, const float *beta, long long E, float eps, float momentum, float *scale,
                       float *shift, float *mean, float *rstd, float *run_mean, float *run_var, long long *nbt,
                       double *sums, void *ws, hipStream_t st);
int gg_att_pairmax_args(const float *Ysrc, const int *nebidx, const float *att16, const float *Wg, const float *b,
                        int B, int Nsrc, int O, const float *Z1, const float *s1, const float *h1, const float *W2,
                        const float *b2, const float *scp, const float *shp, const float *sca, const float *sha,
                        long long ncent, float *agg, int lda, unsigned char *amax, float *zsel, hipStream_t st);
int gg_att_bwd_fused(const GGLinBwd &p, hipStream_t st);       // gridgcn_attbwd.hip; 1 = other shape
int gg_linear_bwd_fused128(const GGLinBwd &p, hipStream_t st); // gridgcn_bwdfused.hip; 1 = other shape
size_t gg_linear_bwd_fused128_workspace(long long E);
void gg_set_bwd_fused128(int v);     // GRIDGCN_OPT_BWD_FUSED128
int gg_get_bwd_fused128();
size_t gg_att_bwd_fused_workspace(long long E, int cin, int C);
int gg_linear_bwd_workspace(long long E, int cin, int C, size_t *bytes, int *nwg);
int gg_linear_bwd(const GGLinBwd &p, hipStream_t st);
int gg_bn_apply(const float *Z, const float *scale, const float *shift, float *Y, long long E,
                int C, int ldy, float drop_p, unsigned long long seed,
                const unsigned long long *seed_dev, hipStream_t st);
void gg_drop_consts(float p, unsigned *thr, float *dscale);
int gg_bn_bwd_reduce(const float *dY, const float *Z, const float *scale, const float *shift,
                     const float *mean, const float *rstd, long long E, int C, double *sums,
                     int ldy, hipStream_t st);
int gg_bn_bwd_elemt(const float *dY, const float *Z, const float *scale, const float *shift,
                    const float *mean, const float *rstd, const float *m1, const float *m2,
                    long long E, int C, float *dZ, hipStream_t st);

// process-wide switch of the register-direct GEMM kernels: 0 = exact fp32 MFMA, 1 = bf16 MFMA with
// fp32 operands in memory, fp32 accumulation and statistics (gridgcn_direct.hip)
void gg_set_att_bwd_fused(int on);   // gridgcn_train.hip (GRIDGCN_OPT_ATT_BWD_FUSED)
int gg_get_att_bwd_fused();
void gg_set_mlp_bf16(int on);
int gg_get_mlp_bf16();
void gg_set_col_split(int on);       // gridgcn_direct.hip (GRIDGCN_OPT_COL_SPLIT)
int gg_get_col_split();
