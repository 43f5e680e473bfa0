// gridgcn_optim.h -- argument table of the one-launch Adam update (gridgcn_optim.hip).
#pragma once
#include <hip/hip_runtime.h>

#define GG_ADAM_MAXT 128      // tensors per launch (the table is a kernel argument: 3.6 KB)
#define GG_ADAM_CHUNK 1024    // elements per workgroup; moment slots are whole chunks

struct GGAdamTable {
    float *p[GG_ADAM_MAXT];
    const float *g[GG_ADAM_MAXT];
    unsigned cstart[GG_ADAM_MAXT + 1];   // first workgroup of tensor i in this launch
    unsigned n[GG_ADAM_MAXT];            // elements
    unsigned mchunk[GG_ADAM_MAXT];       // first chunk of the tensor's slot in the moment buffers
    int nt;
};

// state: int32[2] on the device = (step count t, ticket); zero before the first call
int gg_adam_step(float *const *params, const float *const *grads, const long long *sizes,
                 const long long *mchunk, int n, float *m, float *v, int *state, float lr,
                 const float *lr_dev, float b1, float b2, float eps, float wd, int mode, hipStream_t st);
