// gridgcn_conv.hip -- fused GridConv edge pipeline (gfx950, fp32 MFMA).
//
// Replaces, for one GridConv layer in inference mode, the chain of stock MXNet operators of
//   utils/ops.py:78-93        batch_take_g   (materialises [B,O,P,4+C])
//   gcn_module_g_att.py:190-194,217-218,242-250   geo_vec / geo_dist / att_vec / concat
//   gcn_module_g_att.py:135   pt-MLP   (1x1 conv + BN + ReLU) x npt
//   gcn_module_g_att.py:141,152  att-MLP (10 -> C/4 -> C)
//   gcn_module_g_att.py:167   pair = att * nf
//   gcn_module_g_att.py:57-59 max over the P neighbours (unmasked)
// (>= 12 kernels, each writing a [B,C,O,P] tensor) by ONE kernel that never leaves the CU:
// neighbour rows are gathered from HBM/L2 straight into an LDS tile, every 1x1 conv is an
// fp32 MFMA contraction (v_mfma_f32_32x32x2_f32: exact fp32 FMA chain, 157 TFLOP/s peak) whose
// A operand comes from LDS and whose B operand (BatchNorm-folded weights, L2 resident) comes from
// global memory, activations go back to the same LDS rows, and only [B,O,C] leaves.
//
// Tile: R rows (edges) per workgroup, one wave64 per 32 rows.  Row r of a tile is neighbour
// p = r % P of centre (tile*TO + r / P), TO = R / P centres per tile.
#include "gridgcn_dev.h"
#include "gridgcn_conv.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// C/D layout of v_mfma_f32_32x32x2_f32: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
__device__ __forceinline__ int gg_mfma_row(int reg, int lane) {
    return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}

// acc[nt] (32 rows x 32 cols each) += A[rows, 0:K] * W[0:K, nt*32 : nt*32+32]
// A: LDS, row stride lda (odd -> conflict-free); W: global, row stride ldw (multiple of 32).
template <int NT>
__device__ __forceinline__ void gg_mma_rows(const float *__restrict__ A, int lda,
                                            const float *__restrict__ W, int ldw, int K,
                                            int n0, f32x16 (&acc)[NT])
{
    const int lane = threadIdx.x & 63;
    const float *ap = A + (lane & 31) * lda + (lane >> 5);
    const float *wp = W + (size_t)(lane >> 5) * ldw + n0 + (lane & 31);
    for (int k = 0; k < K; k += 2) {
        float a = ap[k];
        float b[NT];
#pragma unroll
        for (int nt = 0; nt < NT; nt++) b[nt] = wp[(size_t)k * ldw + nt * 32];
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[nt], acc[nt], 0, 0, 0);
    }
}

// one intermediate layer on this wave's 32 rows, in place: A <- relu(A*W + bias)
template <int NT>
__device__ __forceinline__ void gg_layer_inplace(float *A, int lda, const GGConvLayer &L)
{
    const int lane = threadIdx.x & 63;
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[nt][r] = 0.0f;
    gg_mma_rows<NT>(A, lda, L.W, L.ldw, L.K, 0, acc);
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int col = nt * 32 + (lane & 31);
        const float bias = L.b[col];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float v = acc[nt][r] + bias;
            A[gg_mfma_row(r, lane) * lda + col] = v > 0.0f ? v : 0.0f;
        }
    }
    // the next layer reads K = cout (+1 if odd) columns: keep the pad column zero
    if (L.cout_real & 1)
        if (lane < 32) A[lane * lda + L.cout_real] = 0.0f;
}

__device__ __forceinline__ void gg_layer_dispatch(float *A, int lda, const GGConvLayer &L)
{
    switch (L.ldw / 32) {
    case 1: gg_layer_inplace<1>(A, lda, L); break;
    case 2: gg_layer_inplace<2>(A, lda, L); break;
    case 4: gg_layer_inplace<4>(A, lda, L); break;
    case 8: gg_layer_inplace<8>(A, lda, L); break;
    default: break;  // rejected on the host
    }
}

template <int R>
__global__ __launch_bounds__(R * 2) void gg_k_gridconv(GGConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NW = R / 32;
    float *bufA = lds;                              // [R][lda]   pt-MLP activations
    float *bufT = bufA + R * p.lda;                 // [R][ldt]   att input (10) / att hidden
    float *bufR = bufT + R * p.ldt;                 // [R][33]    last-layer tile for the max
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int TO = R / p.P;                         // centres per tile
    const long long c0 = (long long)blockIdx.x * TO;
    const long long ncent = (long long)p.B * p.O;
    const long long nrows_src = (long long)p.B * p.Nsrc;

    // ---- gather + geometric features (batch_take_g + gcn_module_g_att.py:190-250) ----
    for (int r = wave; r < R; r += NW) {
        const int c = r / p.P, pp = r - c * p.P;
        const long long ci = c0 + c;
        float *arow = bufA + r * p.lda;
        float *trow = bufT + r * p.ldt;
        if (c >= TO || ci >= ncent) {
            for (int j = lane; j < p.lda; j += 64) arow[j] = 0.0f;
            for (int j = lane; j < p.ldt; j += 64) trow[j] = 0.0f;
            continue;
        }
        const int b = (int)(ci / p.O);
        long long flat = (long long)p.nebidx[ci * p.P + pp] + (long long)b * p.Nsrc;
        flat = flat < 0 ? 0 : (flat > nrows_src - 1 ? nrows_src - 1 : flat);  // take mode='clip'
        const float *srow = p.src + flat * p.Cs;
        const float *cen = p.cent + ci * p.cent_stride;
        const float cx = cen[0], cy = cen[1], cz = cen[2];
        const float nx = srow[0], ny = srow[1], nz = srow[2];
        const float gx = nx - cx, gy = ny - cy, gz = nz - cz;
        int fo = 0;                                  // first feature column in A
        if (!p.has_feats || p.localfdim != 0) {
            if (lane == 0) { arow[0] = gx; arow[1] = gy; arow[2] = gz; }
            fo = 3;
        }
        if (p.has_feats)
            for (int j = lane; j < p.Cs - 4; j += 64) arow[fo + j] = srow[4 + j];
        const int cin = fo + (p.has_feats ? p.Cs - 4 : 0);
        for (int j = cin + lane; j < p.lda; j += 64) arow[j] = 0.0f;
        if (lane == 0) {
            trow[0] = sqrtf((gx * gx + gy * gy) + gz * gz);
            trow[1] = gx; trow[2] = gy; trow[3] = gz;
            trow[4] = cx; trow[5] = cy; trow[6] = cz;
            trow[7] = nx; trow[8] = ny; trow[9] = nz;
        }
        for (int j = 10 + lane; j < p.ldt; j += 64) trow[j] = 0.0f;
    }
    __syncthreads();

    // ---- per-wave: intermediate pt layers and the first att layer, in place ----
    float *Aw = bufA + wave * 32 * p.lda;
    float *Tw = bufT + wave * 32 * p.ldt;
    for (int l = 0; l < p.npt - 1; l++) gg_layer_dispatch(Aw, p.lda, p.pt[l]);
    gg_layer_dispatch(Tw, p.ldt, p.att[0]);

    // ---- last pt layer x last att layer, 32 columns at a time, max over the P rows ----
    const GGConvLayer &LP = p.pt[p.npt - 1];
    const GGConvLayer &LA = p.att[1];
    const int ntiles = LP.ldw / 32;
    for (int nt = 0; nt < ntiles; nt++) {
        f32x16 accP[1], accA[1];
#pragma unroll
        for (int r = 0; r < 16; r++) { accP[0][r] = 0.0f; accA[0][r] = 0.0f; }
        gg_mma_rows<1>(Aw, p.lda, LP.W, LP.ldw, LP.K, nt * 32, accP);
        gg_mma_rows<1>(Tw, p.ldt, LA.W, LA.ldw, LA.K, nt * 32, accA);
        const int col = nt * 32 + (lane & 31);
        const float bp = LP.b[col], ba = LA.b[col];
        __syncthreads();                             // previous tile's reduction is done
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float vp = accP[0][r] + bp, va = accA[0][r] + ba;
            vp = vp > 0.0f ? vp : 0.0f;
            va = va > 0.0f ? va : 0.0f;
            bufR[(wave * 32 + gg_mfma_row(r, lane)) * 33 + (lane & 31)] = va * vp;   // :167
        }
        __syncthreads();
        for (int t = threadIdx.x; t < TO * 32; t += R * 2) {
            const int c = t >> 5, cc = t & 31;
            const long long ci = c0 + c;
            const int oc = nt * 32 + cc;
            if (ci < ncent && oc < LP.cout_real) {
                const float *rp = bufR + (c * p.P) * 33 + cc;
                float m = rp[0];
                for (int q = 1; q < p.P; q++) m = fmaxf(m, rp[q * 33]);
                p.out[ci * LP.cout_real + oc] = m;
            }
        }
    }
}

int gg_gridconv_forward(const GGConvParams &p, hipStream_t st)
{
    const long long ncent = (long long)p.B * p.O;
    auto launch = [&](auto kern, int R) -> int {
        const int TO = R / p.P;
        if (TO < 1) return 1;
        size_t lds = (size_t)R * (p.lda + p.ldt + 33) * sizeof(float);
        if (lds > 160 * 1024) return 1;
        static bool attr_done[2] = {false, false};
        int slot = R == 128 ? 0 : 1;
        if (!attr_done[slot]) {
            if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024) != hipSuccess) return 3;
            attr_done[slot] = true;
        }
        long long ntile = (ncent + TO - 1) / TO;
        hipLaunchKernelGGL(kern, dim3((unsigned)ntile), dim3(R * 2), lds, st, p);
        return hipGetLastError() == hipSuccess ? 0 : 3;
    };
    // 128-row tiles when they fit in LDS and P allows, else 64-row tiles
    size_t lds128 = (size_t)128 * (p.lda + p.ldt + 33) * sizeof(float);
    if (lds128 <= 160 * 1024 && p.P <= 128) return launch(gg_k_gridconv<128>, 128);
    if (p.P <= 64) return launch(gg_k_gridconv<64>, 64);
    return 1;
}
