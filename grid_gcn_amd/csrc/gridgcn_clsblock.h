// gridgcn_clsblock.h -- host entries of csrc/gridgcn_clsblock.hip (1 = shape not supported).
#pragma once
#include <hip/hip_runtime.h>

int gg_ctx_max(const float *src, const int *nebidx, const float *cent, int cent_stride, int B,
               int Nsrc, int Cs, int O, int P, float *ctx, int *cidx, hipStream_t st);
int gg_ctx_scatter(const float *dctx, const int *cidx, long long ncent, int Cf, int Cs, float *dsrc,
                   hipStream_t st);
int gg_dz_segsum(const float *dY, const float *Z, const float *scale, const float *shift,
                 const float *mean, const float *rstd, const float *m1, const float *m2,
                 long long ncent, int P, int C, float *out, hipStream_t st);
int gg_sparse_add(const unsigned char *amax, const float *gval, long long ncent, int P, int C, float *dX,
                  hipStream_t st);
int gg_bn_stats(const float *Z, long long E, int C, int ld, double *sums, hipStream_t st);
