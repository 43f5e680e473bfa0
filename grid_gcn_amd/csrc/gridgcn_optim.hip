// gridgcn_optim.hip -- the Adam update of every parameter tensor of a model in one launch (gfx950).
//
// Reference: the step's optimizer is mx.optimizer.Adam with wd (segmentation/train_test/base_solver.py:
// 105-114: learning_rate, wd, beta1, beta2).  A segmentation network has 126 parameter tensors of
// 21 .. 32768 floats (340 k in all): the framework's multi-tensor kernel takes four launches of ~20 us
// for what is 10 MB of traffic.  Here the pointer table travels in the kernel arguments (<= 128
// tensors per launch, nothing uploaded, nothing to keep alive for a captured graph), one workgroup owns
// 1024 consecutive elements of one tensor, and the two moment vectors live in flat buffers of whole
// 1024-element chunks.
//
//   g  = grad + wd * w
//   m  = m + (1 - b1) (g - m)          v = b2 v + (1 - b2) g g
//   mode 0 (torch.optim.Adam):  w -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
//   mode 1 (mx.optimizer.Adam): w -= lr sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps)
//
// t lives on the device (replays of a captured step advance it): every workgroup reads it, the last
// one to finish (ticket) writes t + 1 back when `bump` is set (the last launch of a step).
#include <hip/hip_runtime.h>

#include "gridgcn_optim.h"

__global__ __launch_bounds__(256) void gg_k_adam(const GGAdamTable tb, float *__restrict__ m,
                                                 float *__restrict__ v, int *__restrict__ state,
                                                 float lr, const float *__restrict__ lr_dev, float b1,
                                                 float b2, float eps, float wd, int mode, int bump)
{
    const unsigned c = blockIdx.x;
    int lo = 0, hi = tb.nt;                       // tensor of this chunk: cstart[lo] <= c < cstart[lo + 1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tb.cstart[mid] <= c) lo = mid; else hi = mid;
    }
    const unsigned k = c - tb.cstart[lo];         // chunk inside the tensor
    const unsigned n = tb.n[lo];
    float *__restrict__ w = tb.p[lo] + (size_t)k * GG_ADAM_CHUNK;
    const float *__restrict__ g = tb.g[lo] + (size_t)k * GG_ADAM_CHUNK;
    const size_t mo = ((size_t)tb.mchunk[lo] + k) * GG_ADAM_CHUNK;
    const unsigned left = n - k * GG_ADAM_CHUNK;  // elements of the tensor from this chunk on
    const int t = state[0] + 1;
    const double p1 = exp((double)t * log((double)b1)), p2 = exp((double)t * log((double)b2));
    const float bc1 = (float)(1.0 - p1), bc2s = (float)sqrt(1.0 - p2);
    if (lr_dev) lr = *lr_dev;
    const float step = mode ? lr * bc2s / bc1 : lr / bc1;
    const float rs = mode ? 1.f : 1.f / bc2s;
    const unsigned i = threadIdx.x * 4;
    const bool vec = ((((size_t)w | (size_t)g) & 15) == 0) && i + 4 <= left;
    if (vec) {
        float4 W = *(const float4 *)(w + i), G = *(const float4 *)(g + i);
        float4 M = *(const float4 *)(m + mo + i), V = *(const float4 *)(v + mo + i);
        float *Wf = &W.x, *Gf = &G.x, *Mf = &M.x, *Vf = &V.x;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float gg = Gf[j] + wd * Wf[j];
            Mf[j] = Mf[j] + (1.f - b1) * (gg - Mf[j]);
            Vf[j] = b2 * Vf[j] + (1.f - b2) * gg * gg;
            Wf[j] -= step * Mf[j] / (sqrtf(Vf[j]) * rs + eps);
        }
        *(float4 *)(w + i) = W;
        *(float4 *)(m + mo + i) = M;
        *(float4 *)(v + mo + i) = V;
    } else {
        for (unsigned j = i; j < i + 4 && j < left; j++) {
            const float gg = g[j] + wd * w[j];
            const float mm = m[mo + j] + (1.f - b1) * (gg - m[mo + j]);
            const float vv = b2 * v[mo + j] + (1.f - b2) * gg * gg;
            m[mo + j] = mm;
            v[mo + j] = vv;
            w[j] -= step * mm / (sqrtf(vv) * rs + eps);
        }
    }
    __syncthreads();                              // every thread of the workgroup has read state[0]
    if (threadIdx.x == 0) {
        // (relaxed: the ticket publishes no data, it only orders the reads of state[0] above -- complete at
        //  the barrier -- before the one write below; an agent-scope release / acquire here would be an L2
        //  write-back + invalidate per workgroup on this 8-XCD part)
        const int done = __hip_atomic_fetch_add(&state[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (int)gridDim.x - 1) {
            state[1] = 0;
            if (bump) state[0] = t;
        }
    }
}

int gg_adam_step(float *const *params, const float *const *grads, const long long *sizes,
                 const long long *mchunk, int n, float *m, float *v, int *state, float lr,
                 const float *lr_dev, float b1, float b2, float eps, float wd, int mode, hipStream_t st)
{
    for (int i0 = 0; i0 < n || i0 == 0; i0 += GG_ADAM_MAXT) {
        GGAdamTable tb;
        const int nt = n - i0 < GG_ADAM_MAXT ? n - i0 : GG_ADAM_MAXT;
        unsigned chunks = 0;
        for (int i = 0; i < nt; i++) {
            tb.p[i] = params[i0 + i];
            tb.g[i] = grads[i0 + i];
            tb.n[i] = (unsigned)sizes[i0 + i];
            tb.mchunk[i] = (unsigned)mchunk[i0 + i];
            tb.cstart[i] = chunks;
            chunks += (unsigned)((sizes[i0 + i] + GG_ADAM_CHUNK - 1) / GG_ADAM_CHUNK);
        }
        for (int i = nt; i <= GG_ADAM_MAXT; i++) tb.cstart[i] = chunks;
        tb.nt = nt > 0 ? nt : 1;
        const int last = i0 + GG_ADAM_MAXT >= n;
        if (chunks == 0) {
            if (!last) continue;
            // a step in which no parameter has a gradient still counts
            tb.n[0] = 0; tb.mchunk[0] = 0; tb.p[0] = nullptr; tb.g[0] = nullptr;
            tb.cstart[1] = 1;
            chunks = 1;
        }
        hipLaunchKernelGGL(gg_k_adam, dim3(chunks), dim3(256), 0, st, tb, m, v, state, lr, lr_dev, b1, b2,
                           eps, wd, mode, last);
        if (last) break;
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// ---- concat / mask glue of the layer boundary ----------------------------------------------------
// out[r, 0:ca] = a[r, :], out[r, ca:ca+cb] = b[r, :] * mask[r], out[r, ca+cb:ldo] = 0:
//   data_layer = concat(cent, centre features * centmsk)  (segmentation/models/ggcn_models_g.py:186,
//   gcn_module_g_att.py:284-285) in one pass instead of a multiply and a concat;
//   b == nullptr: the cb columns are 1.0 (data = concat(xyz, ones), ggcn_models_g.py:137);
//   out2 (optional): the same rows once more with stride ldo2 -- the zero-padded copy (a multiple of 8
//   floats per row) the centre MLP of the up path reads, which used to cost a fill and a copy.
template <typename I>
__global__ __launch_bounds__(256) void gg_k_cat_mask(const float *__restrict__ a, int lda, int ca,
                                                     const float *__restrict__ b, int ldb, int cb,
                                                     const float *__restrict__ mask,
                                                     float *__restrict__ out, int ldo,
                                                     float *__restrict__ out2, int ldo2, long long total)
{
    // (I = unsigned when the element count allows it: the 64-bit divide per element is half the time of
    //  the [655360, 4 + 8] first level)
    const I ldt = (I)(ldo + ldo2), tot = (I)total;
    for (I i = (I)blockIdx.x * 256 + threadIdx.x; i < tot; i += (I)gridDim.x * 256) {
        const I r = i / ldt;
        int c = (int)(i - r * ldt);
        float *dst = out + (size_t)r * ldo + c;
        if (c >= ldo) { c -= ldo; dst = out2 + (size_t)r * ldo2 + c; }
        float v = 0.f;
        if (c < ca) v = a[(size_t)r * lda + c];
        else if (c < ca + cb) v = b ? b[(size_t)r * ldb + (c - ca)] * (mask ? mask[r] : 1.f) : 1.f;
        *dst = v;
    }
}

// 16-byte form: every width, stride and pointer a multiple of four floats (the layer boundaries of the shipped
// nets: 4 + C columns); one float4 of either output per thread, no per-element division
__global__ __launch_bounds__(256) void gg_k_cat_mask4(const float *__restrict__ a, int lda, int ca,
                                                      const float *__restrict__ b, int ldb, int cb,
                                                      const float *__restrict__ mask,
                                                      float *__restrict__ out, int ldo,
                                                      float *__restrict__ out2, int ldo2, unsigned total4)
{
    const unsigned ldt4 = (unsigned)(ldo + ldo2) >> 2;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total4; i += gridDim.x * 256) {
        const unsigned r = i / ldt4;
        int c = (int)(i - r * ldt4) * 4;
        float *dst = out + (size_t)r * ldo + c;
        if (c >= ldo) { c -= ldo; dst = out2 + (size_t)r * ldo2 + c; }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < ca) v = *(const float4 *)(a + (size_t)r * lda + c);
        else if (c < ca + cb) {
            v = *(const float4 *)(b + (size_t)r * ldb + (c - ca));
            if (mask) { const float m = mask[r]; v.x *= m; v.y *= m; v.z *= m; v.w *= m; }
        }
        *(float4 *)dst = v;
    }
}

// data = concat(xyz, 1) (+ its 8-float padded copy): one thread per point
__global__ __launch_bounds__(256) void gg_k_xyz1(const float *__restrict__ a, int lda, float *__restrict__ out,
                                                 float *__restrict__ out2, long long E)
{
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= E) return;
    const float4 v = make_float4(a[r * lda], a[r * lda + 1], a[r * lda + 2], 1.f);
    *(float4 *)(out + r * 4) = v;
    if (out2) {
        *(float4 *)(out2 + r * 8) = v;
        *(float4 *)(out2 + r * 8 + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// backward of the above: out[r, :] = (g1[r, col0:col0+C] + g2[r, col0:col0+C]) * mask[r]
// (g1 / g2: gradients of the two outputs, either may be nullptr)
template <typename I>
__global__ __launch_bounds__(256) void gg_k_mask_sum(const float *__restrict__ g1, int ld1,
                                                     const float *__restrict__ g2, int ld2, int col0, int C,
                                                     const float *__restrict__ mask,
                                                     float *__restrict__ out, long long total)
{
    const I tot = (I)total;
    for (I i = (I)blockIdx.x * 256 + threadIdx.x; i < tot; i += (I)gridDim.x * 256) {
        const I r = i / (I)C;
        const int c = (int)(i - r * (I)C) + col0;
        float v = g1 ? g1[(size_t)r * ld1 + c] : 0.f;
        if (g2) v += g2[(size_t)r * ld2 + c];
        out[i] = mask ? v * mask[r] : v;
    }
}

__global__ __launch_bounds__(256) void gg_k_mask_sum4(const float *__restrict__ g1, int ld1,
                                                      const float *__restrict__ g2, int ld2, int col0, int C,
                                                      const float *__restrict__ mask,
                                                      float *__restrict__ out, unsigned total4)
{
    const unsigned C4 = (unsigned)C >> 2;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total4; i += gridDim.x * 256) {
        const unsigned r = i / C4;
        const int c = (int)(i - r * C4) * 4 + col0;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g1) v = *(const float4 *)(g1 + (size_t)r * ld1 + c);
        if (g2) {
            const float4 u = *(const float4 *)(g2 + (size_t)r * ld2 + c);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        if (mask) { const float m = mask[r]; v.x *= m; v.y *= m; v.z *= m; v.w *= m; }
        *(float4 *)(out + (size_t)i * 4) = v;
    }
}

static bool gg_al16(const void *p) { return ((size_t)p & 15) == 0; }

int gg_cat_mask(const float *a, int lda, int ca, const float *b, int ldb, int cb, const float *mask,
                float *out, int ldo, float *out2, int ldo2, long long E, hipStream_t st)
{
    if (!out2) ldo2 = 0;
    const long long total = E * (ldo + ldo2);
    if (!b && ca == 3 && cb == 1 && ldo == 4 && (!out2 || ldo2 == 8) && gg_al16(out) && gg_al16(out2)) {
        gg_k_xyz1<<<(int)((E + 255) / 256), 256, 0, st>>>(a, lda, out, out2, E);
        return hipGetLastError() == hipSuccess ? 0 : 3;
    }
    if (b && !((lda | ca | ldb | cb | ldo | ldo2) & 3) && gg_al16(a) && gg_al16(b) && gg_al16(out) && gg_al16(out2) &&
        total < (1ll << 33)) {
        const unsigned total4 = (unsigned)(total >> 2);
        const unsigned nb = (total4 + 255) / 256;
        gg_k_cat_mask4<<<(int)(nb < 16384 ? nb : 16384), 256, 0, st>>>(a, lda, ca, b, ldb, cb, mask, out, ldo, out2,
                                                                       ldo2, total4);
        return hipGetLastError() == hipSuccess ? 0 : 3;
    }
    const long long nb = (total + 255) / 256;
    const int grid = (int)(nb < 8192 ? nb : 8192);
    if (total < (1ll << 31))
        gg_k_cat_mask<unsigned><<<grid, 256, 0, st>>>(a, lda, ca, b, ldb, cb, mask, out, ldo, out2, ldo2, total);
    else
        gg_k_cat_mask<long long><<<grid, 256, 0, st>>>(a, lda, ca, b, ldb, cb, mask, out, ldo, out2, ldo2, total);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_mask_sum(const float *g1, int ld1, const float *g2, int ld2, int col0, int C, const float *mask,
                float *out, long long E, hipStream_t st)
{
    const long long total = E * C;
    if (!((C | col0 | (g1 ? ld1 : 0) | (g2 ? ld2 : 0)) & 3) && gg_al16(g1) && gg_al16(g2) && gg_al16(out) &&
        total < (1ll << 33)) {
        const unsigned total4 = (unsigned)(total >> 2);
        const unsigned nb = (total4 + 255) / 256;
        gg_k_mask_sum4<<<(int)(nb < 16384 ? nb : 16384), 256, 0, st>>>(g1, ld1, g2, ld2, col0, C, mask, out, total4);
        return hipGetLastError() == hipSuccess ? 0 : 3;
    }
    const long long nb = (total + 255) / 256;
    const int grid = (int)(nb < 8192 ? nb : 8192);
    if (total < (1ll << 31))
        gg_k_mask_sum<unsigned><<<grid, 256, 0, st>>>(g1, ld1, g2, ld2, col0, C, mask, out, total);
    else
        gg_k_mask_sum<long long><<<grid, 256, 0, st>>>(g1, ld1, g2, ld2, col0, C, mask, out, total);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
