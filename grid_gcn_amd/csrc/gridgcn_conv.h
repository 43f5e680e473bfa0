// gridgcn_conv.h -- parameter blocks of the fused GridConv kernel.
#pragma once
#include <hip/hip_runtime.h>

struct GGConvLayer {
    const float *W;   // [K][ldw] fp32, k-major; rows >= cin and columns >= cout are zero
    const float *b;   // [ldw]    bias (BatchNorm folded), zero padded
    int K;            // contraction length, even (cin rounded up)
    int ldw;          // padded output width: 32, 64, 128 or 256
    int cout_real;    // true output channels
    int pad_;
};

struct GGConvParams {
    const float *src;      // [B, Nsrc, Cs] fp32: x,y,z,w,features
    const int *nebidx;     // [B, O, P]
    const float *cent;     // centre xyz of centre ci at cent + ci*cent_stride
    float *out;            // [B, O, C]
    int cent_stride;
    int B, Nsrc, Cs, O, P;
    int has_feats, localfdim;
    int npt;               // pt-MLP depth (1..4)
    int lda, ldt;          // LDS row strides (odd)
    GGConvLayer pt[4];
    GGConvLayer att[2];
};

int gg_gridconv_forward(const GGConvParams &p, hipStream_t st);
