// gridgcn_head.hip -- segmentation head loss (gfx950).
//
// Reference: SoftmaxOutput(use_ignore=True, ignore_label=0, normalization='valid')
// (segmentation/models/ggcn_models_g.py:41) on the [B*N, num_classes] logits of get_seg_head
// (:30-43): softmax, cross-entropy averaged over the points whose label is not the ignore label.
// The stock framework path costs a log-softmax pass, an NLL reduction done by one thread block and
// their two backward passes over a [655360, 21] tensor (0.5-0.8 ms each on MI355X); here one thread
// owns a row (<= 32 classes, read as 16-byte pieces of a zero-padded row of `ld` floats):
//   gg_k_ce_fwd   lse[row], sum of -log p[label] and the number of counted rows (fp64 atomics)
//   gg_k_ce_bwd   dlogits[row, c] = (softmax - onehot) * g / count, zeros in ignored rows and in the
//                 padding columns (so the padded tensor feeds the MFMA backward kernels as is)
//   gg_k_colsum   column sums of a [E, ld] tensor (bias gradient of the last linear layer)
#include <hip/hip_runtime.h>

// Sums that every workgroup of a launch adds to: atomics on ONE address are served one after the other
// (512 workgroups: ~20 us, as long as these kernels' reads), so the `slotted` entries spread them over 16
// slots on cache lines of their own, and the last workgroup to arrive adds the slots up and forms what the
// framework formed with 1-3 ops of its own (loss = sum / max(count, 1); the fp32 bias gradient).
// Hand-off per MI355X_MICROARCH "inter-workgroup visibility", the {8-byte agent atomics on both sides}
// form: the partial sums ARE agent-scope atomics, the arriving lane drains them (s_waitcnt vmcnt(0)) before
// its relaxed ticket, the last arriver reads them back with agent-scope atomic loads.  (With
// __threadfence() instead -- an L2 write-back + invalidate per workgroup on this 8-XCD part -- the two
// kernels went 22 -> 45 us.)  Two ticket levels: 16 counters on lines of their own, then one.
// tk: 17 lines of 16 doubles, zero at launch.
#define GG_SLOTS 16
__device__ __forceinline__ int gg_slot() { return (int)(blockIdx.x % GG_SLOTS); }
__device__ __forceinline__ bool gg_last_arriver(double *tk)
{
    const int S = (int)gridDim.x < GG_SLOTS ? (int)gridDim.x : GG_SLOTS;
    const int s = gg_slot();
    const int members = ((int)gridDim.x - s + GG_SLOTS - 1) / GG_SLOTS;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (__hip_atomic_fetch_add((int *)(tk + 16 * s), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != members - 1)
        return false;
    return __hip_atomic_fetch_add((int *)(tk + 16 * GG_SLOTS), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
           S - 1;
}

template <int NV>
__device__ __forceinline__ void gg_row_load(const float *__restrict__ p, float (&v)[4 * NV])
{
#pragma unroll
    for (int q = 0; q < NV; q++) {
        const float4 t = ((const float4 *)p)[q];
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
}

template <int NV>
__global__ __launch_bounds__(256) void gg_k_ce_fwd(const float *__restrict__ logits, int ncls,
                                                   const long long *__restrict__ label,
                                                   long long E, int ignore,
                                                   float *__restrict__ lse_out,
                                                   double *__restrict__ acc,
                                                   float *__restrict__ loss_out)
{
    __shared__ float red[2][4];
    float loss = 0.f, cnt = 0.f;
    for (long long row = (long long)blockIdx.x * 256 + threadIdx.x; row < E;
         row += (long long)gridDim.x * 256) {
        float v[4 * NV];
        gg_row_load<NV>(logits + row * (4 * NV), v);
        float m = -__builtin_inff();
#pragma unroll
        for (int c = 0; c < 4 * NV; c++) if (c < ncls) m = fmaxf(m, v[c]);
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 4 * NV; c++) if (c < ncls) s += expf(v[c] - m);
        const float lse = m + logf(s);
        lse_out[row] = lse;
        const long long lab = label[row];
        if (lab != ignore && lab >= 0 && lab < ncls) {
            float xl = 0.f;
#pragma unroll
            for (int c = 0; c < 4 * NV; c++) xl = (c == (int)lab) ? v[c] : xl;
            loss += lse - xl;
            cnt += 1.f;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        loss += __shfl_xor(loss, o, 64);
        cnt += __shfl_xor(cnt, o, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = loss; red[1][wave] = cnt; }
    __syncthreads();
    // both sums leave the workgroup in ONE atomic instruction (two lanes, one 16-byte piece of a
    // line): every workgroup of the launch adds to the same line and those requests are served one
    // after the other.  cfg4 (655 360 rows): 2 x 2560 requests 70 us, 512 requests 22 us
    if (threadIdx.x < 2) {
        const int t = threadIdx.x;
        // (loss_out: the sums go to one of 16 slots)
        atomicAdd(&acc[t + (loss_out ? 16 * gg_slot() : 0)],
                  (double)((red[t][0] + red[t][1]) + (red[t][2] + red[t][3])));
    }
    if (loss_out && threadIdx.x == 0 && gg_last_arriver(acc + 16 * GG_SLOTS + 16)) {
        // acc: [16 slots x 16] partial (sum, count) | [256] sum, [257] count | tickets
        // (all 32 loads requested together, then added in slot order: one memory round trip instead of 32)
        double v0[GG_SLOTS], v1[GG_SLOTS];
#pragma unroll
        for (int k = 0; k < GG_SLOTS; k++) {
            v0[k] = __hip_atomic_load(&acc[16 * k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v1[k] = __hip_atomic_load(&acc[16 * k + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k < GG_SLOTS; k++) { s0 += v0[k]; s1 += v1[k]; }
        acc[16 * GG_SLOTS] = s0;          // (read by the NEXT launches: gridgcn_softmax_ce_bwd)
        acc[16 * GG_SLOTS + 1] = s1;
        loss_out[0] = (float)(s0 / (s1 > 1.0 ? s1 : 1.0));
    }
}

template <int NV>
__global__ __launch_bounds__(256) void gg_k_ce_bwd(const float *__restrict__ logits, int ncls,
                                                   const long long *__restrict__ label,
                                                   long long E, int ignore,
                                                   const float *__restrict__ lse,
                                                   const double *__restrict__ acc,
                                                   const float *__restrict__ gout,
                                                   const float *__restrict__ cw,
                                                   float *__restrict__ dlogits)
{
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= E) return;
    // normalization='valid' with no valid label: MXNet clamps the count to 1 (loss and gradient 0)
    const float coef = (float)((double)gout[0] / (acc[1] > 1.0 ? acc[1] : 1.0));
    float v[4 * NV], d[4 * NV];
    gg_row_load<NV>(logits + row * (4 * NV), v);
    const long long lab = label[row];
    const bool valid = lab != ignore && lab >= 0 && lab < ncls;
    const float l = lse[row];
#pragma unroll
    for (int c = 0; c < 4 * NV; c++) {
        float g = 0.f;
        if (valid && c < ncls) g = (expf(v[c] - l) - ((c == (int)lab) ? 1.f : 0.f)) * coef;
        d[c] = g;
    }
    if (cw) {
        // weighted_gradient (custom_op/weighted_gradient.py:22-26): the row's gradient times
        // max_c [grad_c < 0] * weight_c -- only the label's entry of softmax - onehot is negative
        float f = 0.f;
#pragma unroll
        for (int c = 0; c < 4 * NV; c++)
            if (c < ncls && d[c] < 0.f) f = fmaxf(f, cw[c]);
#pragma unroll
        for (int c = 0; c < 4 * NV; c++) d[c] *= f;
    }
#pragma unroll
    for (int q = 0; q < NV; q++)
        ((float4 *)(dlogits + row * (4 * NV)))[q] =
            make_float4(d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]);
}

template <int NV>
__global__ __launch_bounds__(256) void gg_k_colsum(const float *__restrict__ X, long long E,
                                                   int ncols, double *__restrict__ out,
                                                   float *__restrict__ out32)
{
    __shared__ float red[4][4 * NV];
    __shared__ int s_last;
    float a[4 * NV];
#pragma unroll
    for (int c = 0; c < 4 * NV; c++) a[c] = 0.f;
    for (long long row = (long long)blockIdx.x * 256 + threadIdx.x; row < E;
         row += (long long)gridDim.x * 256) {
        float v[4 * NV];
        gg_row_load<NV>(X + row * (4 * NV), v);
#pragma unroll
        for (int c = 0; c < 4 * NV; c++) a[c] += v[c];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 4 * NV; c++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a[c] += __shfl_xor(a[c], o, 64);
        if (lane == 0) red[wave][c] = a[c];
    }
    __syncthreads();
    if (threadIdx.x < ncols)
        atomicAdd(&out[threadIdx.x + (out32 ? 32 * gg_slot() : 0)],
                  (double)((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x])));
    if (out32) {
        // out: [16 slots x 32] partial sums | tickets; the fp32 totals by the last workgroup to arrive
        // (ncols <= 32: the atomics above were all issued by wave 0, which drains them before its ticket)
        if (threadIdx.x == 0) s_last = gg_last_arriver(out + 32 * GG_SLOTS);
        __syncthreads();
        if (s_last && threadIdx.x < ncols) {
            double v[GG_SLOTS], a = 0.0;           // (the 16 loads in flight together, added in slot order)
#pragma unroll
            for (int k = 0; k < GG_SLOTS; k++)
                v[k] = __hip_atomic_load(&out[32 * k + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < GG_SLOTS; k++) a += v[k];
            out32[threadIdx.x] = (float)a;
        }
    }
}

// logits rows of ld floats (ld in {4,8,...,32}), ncls <= ld
int gg_ce_fwd(const float *logits, int ld, int ncls, const long long *label, long long E, int ignore,
              float *lse, double *acc, float *loss, hipStream_t st)
{
    if (ld < 4 || ld > 32 || (ld & 3) || ncls < 1 || ncls > ld || E < 1) return 1;
    // at most two workgroups per CU, rows in a grid-stride loop (see the note at the atomics)
    const long long nb = (E + 255) / 256;
    const int grid = (int)(nb < 512 ? nb : 512);
    switch (ld / 4) {
#define GG_CASE(n) case n: gg_k_ce_fwd<n><<<grid, 256, 0, st>>>(logits, ncls, label, E, ignore, lse, acc, loss); break;
    GG_CASE(1) GG_CASE(2) GG_CASE(3) GG_CASE(4) GG_CASE(5) GG_CASE(6) GG_CASE(7) GG_CASE(8)
#undef GG_CASE
    }
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_ce_bwd(const float *logits, int ld, int ncls, const long long *label, long long E, int ignore,
              const float *lse, const double *acc, const float *gout, const float *cw,
              float *dlogits, hipStream_t st)
{
    if (ld < 4 || ld > 32 || (ld & 3) || ncls < 1 || ncls > ld || E < 1) return 1;
    const int grid = (int)((E + 255) / 256);
    switch (ld / 4) {
#define GG_CASE(n) case n: gg_k_ce_bwd<n><<<grid, 256, 0, st>>>(logits, ncls, label, E, ignore, lse, acc, gout, cw, dlogits); break;
    GG_CASE(1) GG_CASE(2) GG_CASE(3) GG_CASE(4) GG_CASE(5) GG_CASE(6) GG_CASE(7) GG_CASE(8)
#undef GG_CASE
    }
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_colsum(const float *X, long long E, int ld, int ncols, double *out, float *out32, hipStream_t st)
{
    if (ld < 4 || ld > 32 || (ld & 3) || ncols < 1 || ncols > ld || E < 1) return 1;
    long long nb = (E + 1023) / 1024;
    const int grid = (int)(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb));
    switch (ld / 4) {
#define GG_CASE(n) case n: gg_k_colsum<n><<<grid, 256, 0, st>>>(X, E, ncols, out, out32); break;
    GG_CASE(1) GG_CASE(2) GG_CASE(3) GG_CASE(4) GG_CASE(5) GG_CASE(6) GG_CASE(7) GG_CASE(8)
#undef GG_CASE
    }
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
