// gridgcn_atteval.h -- parameter block / host entry of the evaluation-mode attention + max kernel.
#pragma once
#include <hip/hip_runtime.h>

struct GGAttEval {
    const float *Z1;           // [E][32] raw output of the first attention conv
    const float *s1, *h1;      // [32] its BatchNorm as scale / shift (running statistics)
    const float *W2, *b2;      // [C][32], [C] second attention conv
    const float *sa, *ha;      // [C] its BatchNorm
    const float *Ysrc;         // [B*Nsrc][C] first point conv applied to the source points
    const int *nebidx;         // [B][O*P]
    const float *att16;        // [E][16]: (dist, gx, gy, gz, ...)
    const float *Wg;           // [3][C] geo_vec weights of the point conv (nullptr: none)
    const float *bp;           // [C] its bias
    const float *sp, *hp;      // [C] BatchNorm of the point conv
    float *out;                // [B*O][ldo]
    long long E;
    int P, O, Nsrc, B, ldo;
};

int gg_att_max_eval(const GGAttEval &p, int C, hipStream_t st);   // 1 = shape not supported
