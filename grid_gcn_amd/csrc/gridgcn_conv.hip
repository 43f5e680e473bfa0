// gridgcn_conv.hip -- fused GridConv edge pipeline (gfx950, fp32 MFMA).
//
// Replaces, for one GridConv layer in inference mode, the chain of stock MXNet operators of
//   utils/ops.py:78-93        batch_take_g   (materialises [B,O,P,4+C])
//   gcn_module_g_att.py:190-194,217-218,242-250   geo_vec / geo_dist / att_vec / concat
//   gcn_module_g_att.py:135   pt-MLP   (1x1 conv + BN + ReLU) x npt
//   gcn_module_g_att.py:141,152  att-MLP (10 -> C/4 -> C)
//   gcn_module_g_att.py:167   pair = att * nf
//   gcn_module_g_att.py:57-59 max over the P neighbours (unmasked)
// (>= 12 kernels, each writing a [B,C,O,P] tensor) by ONE kernel that never leaves the CU.
//
// Work unit = one wave64 = 32 edges (rows):
//   * the 32 neighbour rows are gathered from HBM/L2 straight into the wave's LDS tile, two rows
//     per wave instruction (one per half-wave), several rows in flight;
//   * every 1x1 conv is an fp32 MFMA contraction (v_mfma_f32_32x32x2_f32: exact fp32 FMA chain,
//     157 TFLOP/s peak): A operand from LDS (row stride odd -> conflict-free), B operand =
//     BatchNorm-folded weights read from global/L2 with ONE vector load per k-step (host packs
//     W as [K][32 lanes][NT]) two k-steps ahead of the MFMAs that consume it;
//   * activations return to the same LDS rows; the last pt layer and the last att layer stay in
//     registers, are multiplied, and the max over the P rows of a centre is taken in registers
//     (C/D layout: lane = column, regs = rows) -- only [B,O,C] leaves the CU.
// Waves never synchronise with each other unless a centre spans several waves (P > 32).
#include "gridgcn_dev.h"
#include "gridgcn_once.h"
#include "gridgcn_conv.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NT> struct GGVec;
template <> struct GGVec<1> { typedef float T; };
template <> struct GGVec<2> { typedef float2 T; };
template <> struct GGVec<4> { typedef float4 T; };

template <int NT>
__device__ __forceinline__ float gg_vget(const typename GGVec<NT>::T &v, int i);
template <> __device__ __forceinline__ float gg_vget<1>(const float &v, int) { return v; }
template <> __device__ __forceinline__ float gg_vget<2>(const float2 &v, int i) {
    return i == 0 ? v.x : v.y;
}
template <> __device__ __forceinline__ float gg_vget<4>(const float4 &v, int i) {
    return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}

// C/D layout of v_mfma_f32_32x32x2_f32: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
__device__ __forceinline__ int gg_mfma_row(int reg, int lane) {
    return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}

// acc[nt] (32 rows x 32 cols) += A[32 rows, 0:K] * Wg[0:K, nt*32 : nt*32+32],  nt < NT
// A : LDS, row stride lda (odd).  Wg: global, packed [K][32][NT].  K is a multiple of 4.
template <int NT>
__device__ __forceinline__ void gg_mma(const float *A, int lda, const float *__restrict__ Wg,
                                       int K, f32x16 (&acc)[NT])
{
    typedef typename GGVec<NT>::T V;
    const int lane = threadIdx.x & 63;
    const float *ap = A + (lane & 31) * lda + (lane >> 5);
    const V *wp = (const V *)Wg + ((lane >> 5) * 32 + (lane & 31));
    GG_LOCKSTEP();      // (the rows of A were written by other lanes of this wave: the gather, the layer in front)
    // k-step s covers k = 2s, 2s+1 (lane half selects which); packed row stride = 32 V per k
    float a0 = ap[0], a1 = ap[2];
    V b0 = wp[0], b1 = wp[2 * 32];
    const int nk = K >> 1;  // even
    for (int s = 0; s < nk; s += 2) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, gg_vget<NT>(b0, nt), acc[nt], 0, 0, 0);
        if (s + 2 < nk) { a0 = ap[2 * (s + 2)]; b0 = wp[(size_t)(2 * (s + 2)) * 32]; }
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, gg_vget<NT>(b1, nt), acc[nt], 0, 0, 0);
        if (s + 3 < nk) { a1 = ap[2 * (s + 3)]; b1 = wp[(size_t)(2 * (s + 3)) * 32]; }
    }
}

template <int NT>
__device__ __forceinline__ void gg_zero(f32x16 (&acc)[NT])
{
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[nt][r] = 0.0f;
}

// one intermediate layer on this wave's 32 rows, in place: A <- relu(A*W + bias); width 32*NT
template <int NT>
__device__ __forceinline__ void gg_layer_inplace(float *A, int lda, const GGConvLayer &L)
{
    const int lane = threadIdx.x & 63;
    f32x16 acc[NT];
    gg_zero<NT>(acc);
    gg_mma<NT>(A, lda, L.W, L.K, acc);
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
}

__device__ __forceinline__ void gg_layer_dispatch(float *A, int lda, const GGConvLayer &L)
{
    switch (L.ldw) {
    case 32: gg_layer_inplace<1>(A, lda, L); break;
    case 64: gg_layer_inplace<2>(A, lda, L); break;
    case 128: gg_layer_inplace<4>(A, lda, L); break;
    default: break;  // rejected on the host
    }
}

// last pt layer x last att layer for NT column tiles starting at column n0, then max over rows
template <int NT>
__device__ __forceinline__ void gg_final(const float *Aw, const float *Tw, const GGConvParams &p,
                                         int n0, int wave_in_centre, long long ci_base, int TOw,
                                         float *xbuf)
{
    const GGConvLayer &LP = p.pt[p.npt - 1];
    const GGConvLayer &LA = p.att[1];
    const int lane = threadIdx.x & 63;
    const long long ncent = (long long)p.B * p.O;
    f32x16 accP[NT], accA[NT];
    gg_zero<NT>(accP);
    gg_zero<NT>(accA);
    // packed weights: column group of 32*NT columns starting at n0 -> offset K*n0 floats
    gg_mma<NT>(Aw, p.lda, LP.W + (size_t)LP.K * n0, LP.K, accP);
    gg_mma<NT>(Tw, p.ldt, LA.W + (size_t)LA.K * n0, LA.K, accA);
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int col = n0 + nt * 32 + (lane & 31);
        const float bp = LP.b[col], ba = LA.b[col];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float vp = accP[nt][r] + bp, va = accA[nt][r] + ba;
            vp = vp > 0.0f ? vp : 0.0f;
            va = va > 0.0f ? va : 0.0f;
            accP[nt][r] = va * vp;                       // pair = att * nf (:167)
        }
    }
    const float NEG = -__builtin_inff();
    if (p.P <= 32) {
        // the wave holds TOw whole centres: rows [c*P, (c+1)*P)
        for (int c = 0; c < TOw; c++) {
            const int lo = c * p.P, hi = lo + p.P;
            const long long ci = ci_base + c;
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                float m = NEG;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    int row = gg_mfma_row(r, lane);
                    float v = accP[nt][r];
                    m = (row >= lo && row < hi) ? fmaxf(m, v) : m;
                }
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                const int col = n0 + nt * 32 + (lane & 31);
                if (lane < 32 && ci < ncent && col < LP.cout_real) p.out[ci * LP.cout_real + col] = m;
            }
        }
    } else {
        // the centre spans several waves: rows of this wave are p = wave_in_centre*32 + row
        const int nrow = p.P - wave_in_centre * 32;      // valid rows in this wave (may be > 32)
        const int nwv = (p.P + 31) >> 5;
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            float m = NEG;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int row = gg_mfma_row(r, lane);
                m = (row < nrow) ? fmaxf(m, accP[nt][r]) : m;
            }
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            if (lane < 32) xbuf[wave_in_centre * 128 + nt * 32 + lane] = m;
        }
        __syncthreads();
        if (wave_in_centre == 0 && ci_base < ncent) {
            for (int j = lane; j < NT * 32; j += 64) {
                float m = xbuf[j];
                for (int w = 1; w < nwv; w++) m = fmaxf(m, xbuf[w * 128 + j]);
                const int col = n0 + j;
                if (col < LP.cout_real) p.out[ci_base * LP.cout_real + col] = m;
            }
        }
        __syncthreads();
    }
}

// blockDim = 64 * WPC (WPC = waves per centre = ceil(P/32) when P > 32, else 1)
// (amdgpu_waves_per_eu(2): 256 registers a lane -- and the MFMA accumulators in ordinary VGPRs; with the default bound the
//  compiler kept them in AGPRs: 1344 v_accvgpr copies in the loop, found at the end of round 5)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gg_k_gridconv(GGConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nw = blockDim.x >> 6;
    float *Aw = lds + (size_t)wave * 32 * (p.lda + p.ldt);   // [32][lda]
    float *Tw = Aw + 32 * p.lda;                              // [32][ldt]
    float *xbuf = lds + (size_t)nw * 32 * (p.lda + p.ldt);    // [nw][128] cross-wave max
    const long long ncent = (long long)p.B * p.O;
    const long long nrows_src = (long long)p.B * p.Nsrc;
    const int TOw = p.P <= 32 ? 32 / p.P : 1;                 // whole centres per wave
    const long long ci_base = p.P <= 32 ? ((long long)blockIdx.x * nw + wave) * TOw
                                        : (long long)blockIdx.x;
    const int prow0 = p.P <= 32 ? 0 : wave * 32;              // neighbour number of row 0

    // ---- row descriptors: lane r < 32 owns row r of this wave ----
    long long myflat = -1;      // source row (clipped flat index) or -1 for a padding row
    long long myci = -1;
    {
        const int r = lane & 31;
        int c, pp;
        if (p.P <= 32) { c = r / p.P; pp = r - c * p.P; if (c >= TOw) c = -1; }
        else { c = 0; pp = prow0 + r; if (pp >= p.P) c = -1; }
        if (c >= 0 && ci_base + c < ncent) {
            myci = ci_base + c;
            const int b = (int)(myci / p.O);
            long long f = (long long)p.nebidx[myci * p.P + pp] + (long long)b * p.Nsrc;
            myflat = f < 0 ? 0 : (f > nrows_src - 1 ? nrows_src - 1 : f);   // take mode='clip'
        }
    }
    // ---- gather: half-wave h loads row (2*it + h); lane j = float4 j of the row ----
    const int half = lane >> 5, j = lane & 31;
    const int geo = (!p.has_feats || p.localfdim != 0);
    if ((p.Cs & 3) == 0) {
        const int nq = p.Cs >> 2;                             // float4 per row
#pragma unroll 4
        for (int it = 0; it < 16; it++) {
            const int r = 2 * it + half;
            const long long flat = __shfl(myflat, r, 64);
            const long long ci = __shfl(myci, r, 64);
            float *arow = Aw + r * p.lda;
            float *trow = Tw + r * p.ldt;
            if (flat < 0) {
                for (int q = j; q < p.lda; q += 32) arow[q] = 0.0f;
                for (int q = j; q < p.ldt; q += 32) trow[q] = 0.0f;
                continue;
            }
            const float4 *srow = (const float4 *)(p.src + flat * p.Cs);
            for (int q = j; q < nq; q += 32) {
                float4 v = srow[q];
                if (q == 0) {
                    const float *cen = p.cent + ci * p.cent_stride;
                    const float cx = cen[0], cy = cen[1], cz = cen[2];
                    const float gx = v.x - cx, gy = v.y - cy, gz = v.z - cz;
                    trow[0] = sqrtf((gx * gx + gy * gy) + gz * gz);
                    trow[1] = gx; trow[2] = gy; trow[3] = gz;
                    trow[4] = cx; trow[5] = cy; trow[6] = cz;
                    trow[7] = v.x; trow[8] = v.y; trow[9] = v.z;
                    // columns 0..3 of A: geo_vec + a zero pad column (or zeros when the layer
                    // takes no geometry: the matching weight rows are zero as well)
                    v = geo ? make_float4(gx, gy, gz, 0.0f) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                float *d = arow + 4 * q;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
            for (int q = p.Cs + j; q < p.lda; q += 32) arow[q] = 0.0f;
            for (int q = 10 + j; q < p.ldt; q += 32) trow[q] = 0.0f;
        }
    } else {
        for (int it = 0; it < 16; it++) {
            const int r = 2 * it + half;
            const long long flat = __shfl(myflat, r, 64);
            const long long ci = __shfl(myci, r, 64);
            float *arow = Aw + r * p.lda;
            float *trow = Tw + r * p.ldt;
            if (flat < 0) {
                for (int q = j; q < p.lda; q += 32) arow[q] = 0.0f;
                for (int q = j; q < p.ldt; q += 32) trow[q] = 0.0f;
                continue;
            }
            const float *srow = p.src + flat * p.Cs;
            for (int q = 4 + j; q < p.Cs; q += 32) arow[q] = srow[q];
            if (j == 0) {
                const float *cen = p.cent + ci * p.cent_stride;
                const float cx = cen[0], cy = cen[1], cz = cen[2];
                const float nx = srow[0], ny = srow[1], nz = srow[2];
                const float gx = nx - cx, gy = ny - cy, gz = nz - cz;
                trow[0] = sqrtf((gx * gx + gy * gy) + gz * gz);
                trow[1] = gx; trow[2] = gy; trow[3] = gz;
                trow[4] = cx; trow[5] = cy; trow[6] = cz;
                trow[7] = nx; trow[8] = ny; trow[9] = nz;
                arow[0] = geo ? gx : 0.f; arow[1] = geo ? gy : 0.f; arow[2] = geo ? gz : 0.f;
                arow[3] = 0.0f;
            }
            for (int q = p.Cs + j; q < p.lda; q += 32) arow[q] = 0.0f;
            for (int q = 10 + j; q < p.ldt; q += 32) trow[q] = 0.0f;
        }
    }
    // LDS traffic of one wave is ordered; no other wave touches these rows.

    // ---- intermediate pt layers and the first att layer, in place ----
    for (int l = 0; l < p.npt - 1; l++) gg_layer_dispatch(Aw, p.lda, p.pt[l]);
    gg_layer_dispatch(Tw, p.ldt, p.att[0]);

    // ---- last pt layer (x) last att layer, product, max over the P rows ----
    const int ldw = p.pt[p.npt - 1].ldw;
    if (ldw == 32) gg_final<1>(Aw, Tw, p, 0, wave, ci_base, TOw, xbuf);
    else if (ldw == 64) gg_final<2>(Aw, Tw, p, 0, wave, ci_base, TOw, xbuf);
    else
        for (int n0 = 0; n0 < ldw; n0 += 128) gg_final<4>(Aw, Tw, p, n0, wave, ci_base, TOw, xbuf);
}

int gg_gridconv_forward(const GGConvParams &p, hipStream_t st)
{
    const long long ncent = (long long)p.B * p.O;
    static GGDevOnce attr_done;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)gg_k_gridconv,
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess) return 3;
        attr_done = true;
    }
    int nw;
    long long nblk;
    if (p.P <= 32) {
        const int TOw = 32 / p.P;
        const long long nwave = (ncent + TOw - 1) / TOw;
        nw = 1;                                  // independent single-wave workgroups
        nblk = nwave;
    } else {
        nw = (p.P + 31) / 32;                    // 2..4 waves share one centre
        nblk = ncent;
    }
    size_t lds = ((size_t)nw * 32 * (p.lda + p.ldt) + (size_t)nw * 128) * sizeof(float);
    if (lds > 160 * 1024 || nblk > 0x7fffffffLL) return 1;
    hipLaunchKernelGGL(gg_k_gridconv, dim3((unsigned)nblk), dim3(64 * nw), lds, st, p);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
