// gridgcn_mma.h -- fp32 MFMA building blocks shared by the training kernels (gfx950, wave64).
// v_mfma_f32_32x32x2_f32: D[32x32] += A[32x2] * B[2x32]; exact fp32 FMA chain, 157.3 TFLOP/s peak.
//   A operand: lane l holds A[row = l&31][k = l>>5]      (one float)
//   B operand: lane l holds B[k = l>>5][col = l&31]      (one float)
//   C/D      : lane l holds col = l&31, rows (reg&3) + 8*(reg>>2) + 4*(l>>5), reg = 0..15
#pragma once
#include <hip/hip_runtime.h>

typedef float ggm_f32x16 __attribute__((ext_vector_type(16)));

template <int NT> struct GGMVec;
template <> struct GGMVec<1> { typedef float T; };
template <> struct GGMVec<2> { typedef float2 T; };
template <> struct GGMVec<4> { typedef float4 T; };

template <int NT> __device__ __forceinline__ float ggm_vget(const typename GGMVec<NT>::T &v, int i);
template <> __device__ __forceinline__ float ggm_vget<1>(const float &v, int) { return v; }
template <> __device__ __forceinline__ float ggm_vget<2>(const float2 &v, int i) {
    return i == 0 ? v.x : v.y;
}
template <> __device__ __forceinline__ float ggm_vget<4>(const float4 &v, int i) {
    return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}

// Dropout mask as a counter-based hash of (seed, element index): the forward pass and the backward
// pass regenerate the same bit, no mask tensor is stored.  Element idx of a [E][C] activation is
// row*C + col.  keep <=> hash >= thr with thr = p * 2^32 (p = drop probability; thr 0 keeps all).
__device__ __forceinline__ bool gg_drop_keep(unsigned long long idx, unsigned seed_lo,
                                             unsigned seed_hi, unsigned thr)
{
    unsigned h = (unsigned)idx ^ seed_lo;
    h *= 0x9E3779B1u;
    h ^= (unsigned)(idx >> 32) * 0x7FEB352Du + seed_hi;
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h >= thr;
}

__device__ __forceinline__ int ggm_row(int reg, int lane) {
    return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}

template <int NT> __device__ __forceinline__ void ggm_zero(ggm_f32x16 (&acc)[NT])
{
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[nt][r] = 0.0f;
}

// acc[nt] += A[32 rows, 0:K] * Wg[0:K, nt*32 : nt*32+32]
// A : LDS, row-major, row stride lda (odd => conflict-free column reads).
// Wg: global, packed [K][32 lanes][NT] (one vector load per k-step).  K multiple of 4.
// same contraction with the packed weights resident in LDS (persistent workgroups load them once)
template <int NT>
__device__ __forceinline__ void ggm_mma_lds(const float *A, int lda, const float *Wl, int K,
                                            ggm_f32x16 (&acc)[NT])
{
    typedef typename GGMVec<NT>::T V;
    const int lane = threadIdx.x & 63;
    const float *ap = A + (lane & 31) * lda + (lane >> 5);
    const V *wp = (const V *)Wl + ((lane >> 5) * 32 + (lane & 31));
    float a0 = ap[0], a1 = ap[2];
    V b0 = wp[0], b1 = wp[2 * 32];
    const int nk = K >> 1;
    for (int s = 0; s < nk; s += 2) {
        float a2 = 0.f, a3 = 0.f;
        V b2 = b0, b3 = b1;
        if (s + 2 < nk) { a2 = ap[2 * (s + 2)]; b2 = wp[(2 * (s + 2)) * 32]; }
        if (s + 3 < nk) { a3 = ap[2 * (s + 3)]; b3 = wp[(2 * (s + 3)) * 32]; }
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, ggm_vget<NT>(b0, nt), acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, ggm_vget<NT>(b1, nt), acc[nt], 0, 0, 0);
        a0 = a2; a1 = a3; b0 = b2; b1 = b3;
    }
}

template <int NT>
__device__ __forceinline__ void ggm_mma(const float *A, int lda, const float *__restrict__ Wg,
                                        int K, ggm_f32x16 (&acc)[NT])
{
    typedef typename GGMVec<NT>::T V;
    const int lane = threadIdx.x & 63;
    const float *ap = A + (lane & 31) * lda + (lane >> 5);
    const V *wp = (const V *)Wg + ((lane >> 5) * 32 + (lane & 31));
    float a0 = ap[0], a1 = ap[2];
    V b0 = wp[0], b1 = wp[2 * 32];
    const int nk = K >> 1;
    for (int s = 0; s < nk; s += 2) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, ggm_vget<NT>(b0, nt), acc[nt], 0, 0, 0);
        if (s + 2 < nk) { a0 = ap[2 * (s + 2)]; b0 = wp[(size_t)(2 * (s + 2)) * 32]; }
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, ggm_vget<NT>(b1, nt), acc[nt], 0, 0, 0);
        if (s + 3 < nk) { a1 = ap[2 * (s + 3)]; b1 = wp[(size_t)(2 * (s + 3)) * 32]; }
    }
}

// ------------------------------------------------------------------------------------------
// Buffer addressing for loops that walk a row block: the descriptor (four scalar registers) holds a
// wave-uniform base, a lane supplies ONE constant byte offset, the row / column-tile step is a
// scalar offset -- a load or store is one instruction with no per-element address arithmetic
// (global_load/store with 64-bit per-lane pointers costs a v_lshl_add_u64 each, and every
// instruction a wave issues takes ~5 cycles from the MFMA pipe: tools/micro/mfma_valu.hip).
// Raw buffer (stride 0), 32-bit data format, range check off (num_records = 2^32 - 1).
// The LLVM intrinsics are declared directly: clang's __builtin_amdgcn_raw_buffer_load_b64/_b128 of
// this toolchain return the first dword in every element.
typedef int gg_rsrc __attribute__((ext_vector_type(4)));
typedef float gg_f32x2 __attribute__((ext_vector_type(2)));
typedef float gg_f32x4 __attribute__((ext_vector_type(4)));
typedef int gg_i32x4 __attribute__((ext_vector_type(4)));
__device__ gg_f32x4 gg_buf_ld4(gg_rsrc r, unsigned lane_bytes, unsigned uniform_bytes, int aux = 0) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ gg_f32x2 gg_buf_ld2(gg_rsrc r, unsigned lane_bytes, unsigned uniform_bytes, int aux = 0) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
__device__ float gg_buf_ld(gg_rsrc r, unsigned lane_bytes, unsigned uniform_bytes, int aux = 0) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ unsigned char gg_buf_ld_u8(gg_rsrc r, unsigned lane_bytes, unsigned uniform_bytes, int aux = 0) __asm("llvm.amdgcn.raw.buffer.load.i8");
__device__ unsigned short gg_buf_ld_u16(gg_rsrc r, unsigned lane_bytes, unsigned uniform_bytes, int aux = 0) __asm("llvm.amdgcn.raw.buffer.load.i16");
__device__ unsigned gg_buf_ld_u32(gg_rsrc r, unsigned lane_bytes, unsigned uniform_bytes, int aux = 0) __asm("llvm.amdgcn.raw.buffer.load.i32");
__device__ void gg_buf_st(float v, gg_rsrc r, unsigned lane_bytes, unsigned uniform_bytes, int aux = 0) __asm("llvm.amdgcn.raw.buffer.store.f32");
__device__ __forceinline__ gg_rsrc gg_make_rsrc(const void *uniform_base)
{
    const unsigned long long a = (unsigned long long)uniform_base;
    gg_rsrc r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));   // (stride 0: bits 48..61 clear)
    r.z = -1;
    r.w = 0x00020000;
    return r;
}

// the same with a range check: raw buffer, stride 0, num_records in bytes -- loads beyond it return 0, stores beyond it
// are dropped (the partial last tile of a loop needs no predicate)
__device__ __forceinline__ gg_rsrc gg_make_rsrc_n(const void *uniform_base, unsigned bytes)
{
    gg_rsrc r = gg_make_rsrc(uniform_base);
    r.z = (int)bytes;
    return r;
}

// ------------------------------------------------------------------------------------------
// Operand formulas of the training kernels, written for instruction count (every VALU instruction
// costs the SIMD ~5 cycles of MFMA time): explicit FMAs on float4 values, which the compiler issues
// as v_pk_fma_f32 / v_pk_add_f32 (two channels per instruction).
__device__ __forceinline__ gg_f32x4 gg_ld_f4(const float *p) { return *(const gg_f32x4 *)p; }

// BatchNorm + ReLU of four consecutive channels: max(x * sc + sh, 0)
__device__ __forceinline__ gg_f32x4 gg_bnrelu4v(gg_f32x4 x, gg_f32x4 sc, gg_f32x4 sh)
{
    const gg_f32x4 y = __builtin_elementwise_fma(x, sc, sh);
    return __builtin_elementwise_max(y, (gg_f32x4)(0.f));
}

// dZ of four consecutive channels k .. k+3 of one row (BatchNorm + ReLU backward):
//   dz = sc * (mask ? g : 0) + ((z - mu) * bz + cz),   mask = z * sc + sh > 0  [and arg-max byte == pp]
// cst = [5][C]: sc, sh, mu, bz, cz (bz = -sc*rstd*m2, cz = -sc*m1).  am = the four arg-max bytes of the
// sparse upstream gradient (gridgcn_pairmax_bwd), consulted only when `sparse`.
__device__ __forceinline__ gg_f32x4 gg_dz4v(gg_f32x4 z, gg_f32x4 g, unsigned am, int pp, bool sparse,
                                            const float *cst, int C, int k)
{
    const gg_f32x4 sc = gg_ld_f4(cst + k), sh = gg_ld_f4(cst + C + k), mu = gg_ld_f4(cst + 2 * C + k);
    const gg_f32x4 bz = gg_ld_f4(cst + 3 * C + k), cz = gg_ld_f4(cst + 4 * C + k);
    const gg_f32x4 y = __builtin_elementwise_fma(z, sc, sh);
    const gg_f32x4 t = __builtin_elementwise_fma(z - mu, bz, cz);
    // (bit-wise & | on purpose: && || made a branch per element)
    const bool ns = !sparse;
    gg_f32x4 gm;
    gm.x = ((y.x > 0.f) & (ns | ((int)(am & 255u) == pp))) ? g.x : 0.f;
    gm.y = ((y.y > 0.f) & (ns | ((int)((am >> 8) & 255u) == pp))) ? g.y : 0.f;
    gm.z = ((y.z > 0.f) & (ns | ((int)((am >> 16) & 255u) == pp))) ? g.z : 0.f;
    gm.w = ((y.w > 0.f) & (ns | ((int)(am >> 24) == pp))) ? g.w : 0.f;
    return __builtin_elementwise_fma(sc, gm, t);
}

// n float4 from src (every sstride-th) to LDS, eight loads in flight per thread.  (The plain loop -- load, wait, LDS
// store, sixteen times for a 256 x 128 operand -- opened EVERY forward / dX launch with ~10 us in which the whole
// chip waited for L2 round trips one after the other; found in the ISA at the end of round 4.)
__device__ __forceinline__ void gg_stage_copy4(float4 *dst, const float4 *__restrict__ src, int n, int sstride,
                                               int tid, int nthr)
{
    for (int i0 = tid; i0 < n; i0 += nthr * 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + u * nthr;
            v[u] = src[(size_t)(i < n ? i : n - 1) * sstride];
        }
        // (no `if (i < n)` around the stores: the compiler sinks a load whose only use is conditional into that
        //  branch -- load, wait, store, eight times, as the ISA of the first version showed.  Past the end the last
        //  element is stored again, with its own value.)
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + u * nthr;
            dst[i < n ? i : n - 1] = v[u];
        }
    }
}

