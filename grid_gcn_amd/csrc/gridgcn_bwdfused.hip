// gridgcn_bwdfused.hip -- backward of a 128-output conv + BatchNorm + ReLU layer of the per-point stacks in ONE
// pass over its [E, 128] pre-activation and its dense upstream gradient (gfx950, fp32 MFMA; round 5).
//
// The register-direct pair (gridgcn_direct.hip) reads Z and dY twice -- gg_k_linear_dx_direct forms dZ for
// dX = dZ W, gg_k_linear_dw_direct forms it again for dW = dZ^T act(X) -- 0.67 GB of the 1.34 + 1.0 GB the two
// move at fc1 of the 81 920-point net, and neither is bound by its MFMAs (51 % of the matrix pipe together).
// Here a workgroup of 8 waves (two per SIMD, one workgroup per CU: 142 KB of LDS) holds the dX operand W once and
// runs two independent HALVES of four waves, each owning 64-row blocks, a transposed dZ tile and a barrier of its own
// (an LDS counter: gfx950 has no named barriers), and every dZ and every act(X) value is formed exactly ONCE per block:
//
//   (w4 = wave & 3 inside a half; tw = w4 & 1 row tile of the block, hw = w4 >> 1, ct = w4)
//   phase 1   wave (tw, hw): dZ of row tile tw for the channel half hw (lane = row, gg_dz4v as everywhere): A operand
//             of a PARTIAL dX[32 rows, all four column tiles] over those 64 channels (32 steps x 4 MFMAs, W from LDS:
//             one formed value feeds four MFMAs), and parked TRANSPOSED in the block's LDS tile T[tile][channel][row]
//   barrier
//   phase 2   wave ct owns dW[all 128 channels, column tile ct] for the block's two row tiles:
//             B = act(X) read straight from memory in the C/D row order (lane = column; a coalesced 128-byte row piece
//             per half-wave and load) -- one formed value feeds the four channel slices' MFMAs -- A = dZ^T from T
//             (lane = channel, 16 rows: four ds_read_b128)
//   barrier   the wave pair of a row tile swaps, through the (now free) T area, the halves of its partial dX
//   barrier   epilogue: dX = half 0 + half 1 of row tile tw, column tiles 2hw, 2hw + 1; stores; the BatchNorm-backward
//             sums of the layer in front;  barrier (T is the next block's)
//
// 256 MFMAs per wave and block against ~990 other instructions (the first form -- the pair of a row tile splitting
// the COLUMNS, each forming all of dZ, one act(X) value per MFMA -- had 1500 and was slower than the separate
// kernels: at two waves per SIMD this kernel's time is its instruction count).  Loads run ahead of
// their use in rings sized to what 256 registers leave beside 128 accumulators: four (Z, dY) quads, the phase-2
// tiles behind phase 1's last quads, the next block's first quads behind the exchange.  Z / dY / X / dX each cross
// the memory system once (X a second time out of L2 for the epilogue's 16 KB per wave).
// Results: same terms as the separate kernels in other summation orders (tests: dX 2e-6, dW 1e-5, sums 1e-6 of the
// largest entry; both against float64).  128 input columns per launch ([dx_col0, +128) of a layer with cin = 128 or
// 256: the 256-wide update conv takes two launches, forming dZ twice); E % 128 == 0, dense dY, fp32.
#include "gridgcn_mma.h"
#include "gridgcn_once.h"
#include "gridgcn_train.h"

#define GG_BF_TS 36          // LDS stride (floats) of one channel's 32 rows of a dZ tile: 32 + 4
#define GG_BF_C 128

static int g_bwd_fused128 = 1;          // GRIDGCN_OPT_BWD_FUSED128
void gg_set_bwd_fused128(int v) { g_bwd_fused128 = v < 0 ? 0 : (v > 2 ? 2 : v); }   // 0 off, 1 all shapes, 2 cin = 128 only
int gg_get_bwd_fused128() { return g_bwd_fused128; }

__device__ __forceinline__ float gg_bf_f4(const gg_f32x4 &v, int i)
{
    return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}

__global__ __launch_bounds__(512) void gg_k_linear_bwd_fused128(GGLinBwd p, float *__restrict__ part)
{
    constexpr int C = GG_BF_C;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    // Two independent HALVES of four waves (waves 0-3, 4-7) share the workgroup's LDS operand but own separate 64-row
    // blocks, transposed tiles and barriers (gg_half_barrier: an LDS counter), so that one half's barrier / epilogue
    // stalls overlap the other half's MFMAs -- with one 8-wave team both waves of every SIMD sat at the same barrier.
    // Inside a half: w4 = wave & 3;  phase 1 / epilogue: row tile tw = w4 & 1, channel half hw = w4 >> 1;
    // phase 2: column tile ct = w4, both row tiles of the block.
    const int half = wave >> 2, w4 = wave & 3;
    const int tw = w4 & 1, hw = w4 >> 1, ct = w4;
    // (the per-channel constants FIRST: a ds_read reaches 64 KB from its base register by immediate offset -- behind
    //  the 64-KB operand every constant array of every quad got an address register of its own, 33 of them spilled,
    //  and a scratch reload waits for ALL outstanding loads: vmcnt(0) in front of every MFMA group)
    float *cst = lds;                             // scale, shift, mean, bz, cz  [5][C]
    float *pcs = cst + 5 * C;                     // layer in front: scale, shift, mean, rstd  [4][128]
    float *T = pcs + 4 * 128;                     // [2 halves][2 row tiles][C][TS]; after phase 2: the dX exchange [4 waves][32][64]
    float *Wl = T + 4 * C * GG_BF_TS;             // [64 steps][64 lanes][4 column tiles]
    const int ldx = p.cin, col0 = p.dx_col0;
    const int ldz = p.ldz ? p.ldz : C;
    {
        gg_stage_copy4((float4 *)Wl, (const float4 *)p.Wdx + (col0 ? 1 : 0), 64 * 64, p.dx_wstride, tid, 512);
        for (int c = tid; c < C; c += 512) {
            const float sc = p.scale[c];
            cst[c] = sc;
            cst[C + c] = p.shift[c];
            cst[2 * C + c] = p.mean[c];
            float m1v, m2v;
            gg_bn_m12(p, c, m1v, m2v);
            cst[3 * C + c] = -(sc * p.rstd[c]) * m2v;
            cst[4 * C + c] = -(sc * m1v);
        }
        const bool pb = p.pscale != nullptr;
        for (int c = tid; c < 128; c += 512) {
            const int col = col0 + c;
            // (no BatchNorm in front: act(x) = x -- scale 1, shift 0, and the ReLU floor below is -inf)
            pcs[c] = pb ? p.pscale[col] : 1.f;
            pcs[128 + c] = pb ? p.pshift[col] : 0.f;
            pcs[256 + c] = pb ? p.pmean[col] : 0.f;
            pcs[384 + c] = pb ? p.prstd[col] : 0.f;
        }
    }
    if (tid < 2) ((int *)(Wl + 64 * 64 * 4))[tid] = 0;
    __syncthreads();
    const bool prevbn = p.pscale != nullptr;
    const float lo = prevbn ? 0.f : -__builtin_inff();
    // phase 2 role: column tile tw, row tiles 2hw and 2hw + 1; epilogue role: row tile tw, column tiles 2hw and
    // 2hw + 1.  (The per-column constants of the layer in front are re-read from LDS where they are used: kept in
    // registers across the block they were what spilled.)
    bool tbn[2];
    unsigned lcd[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        tbn[t] = prevbn && (p.nbn == 0 || col0 + (2 * hw + t) * 32 < p.nbn);     // (wave uniform)
        // lane part of an address in a 32-row x 128-column block of [.][ldx] floats, C/D order: row 4h, column of tile
        lcd[t] = (unsigned)(4 * h * ldx + (2 * hw + t) * 32 + l31) * 4u;
    }
    const unsigned lcb = (unsigned)(4 * h * ldx + ct * 32 + l31) * 4u;
    float a1[2] = {0.f, 0.f}, a2[2] = {0.f, 0.f};
    ggm_f32x16 accw[4];
    ggm_zero<4>(accw);
    const long long nblk = p.E >> 6;              // 64-row blocks: half h takes blocks 2 * blockIdx.x + h, + 2 * gridDim.x, ..
    float *Th = T + (size_t)half * (2 * C * GG_BF_TS);
    float *Tw = Th + (size_t)tw * (C * GG_BF_TS);
    float *Xw = Th + (size_t)w4 * 2048, *Xp = Th + (size_t)(w4 ^ 2) * 2048;   // dX exchange: mine, my partner's
    // barrier of the four waves of a half: arrivals counted in LDS (the writes in front of it drained first), then a
    // poll with s_sleep; `bar_target` advances by 4 per barrier
    int *bar = (int *)(Wl + 64 * 64 * 4) + half;
    int bar_target = 0;
    auto half_barrier = [&]() {
        bar_target += 4;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < bar_target)
            __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
    };
    // Phase 1's (Z, dY) stream as a ring of four QUADS (4 channels of Z and of dY per lane each: 8 registers): quad
    // q + 4 is requested when quad q has been consumed, i.e. three quads = 48 MFMAs ahead of its use; the first four
    // of a block are requested during the block in front.  (Whole 32-channel chunks, two of them resident, did not
    // fit beside the 128 accumulator registers: 80 spills, and a scratch reload waits for ALL loads in flight.)
    struct Quad { gg_f32x4 z, g; };
    Quad Q[4];
    const int kw = hw * 64 + h * 16;                       // first channel of this lane in the wave's channel half
    auto ldq = [&](Quad &q_, long long row0, int qi) {
        const int k = kw + (qi >> 2) * 32 + (qi & 3) * 4;
        q_.z = gg_ld_f4(p.Z + (row0 + l31) * ldz + k);
        q_.g = gg_ld_f4(p.dY + (row0 + l31) * p.ldy + k);
    };
    const long long blk0 = 2 * (long long)blockIdx.x + half, bstep = 2 * (long long)gridDim.x;
    if (blk0 < nblk) {
#pragma unroll
        for (int qi = 0; qi < 4; qi++) ldq(Q[qi], (blk0 << 6) + tw * 32, qi);
    }
    // (a start offset of half a block between the halves -- s_sleep in front of the second one's loop -- measured: no
    //  change; the halves drift apart by themselves)
    for (long long blk = blk0; blk < nblk; blk += bstep) {
        const long long b0 = blk << 6, r0 = b0 + tw * 32;
        // 16 rows (C/D order) of one column tile of a 32-row tile of X
        auto ldx16 = [&](float (&x)[16], long long row0, unsigned lanepart) {
            const gg_rsrc rx = gg_make_rsrc(p.Aprev + (row0 * ldx + col0));
#pragma unroll
            for (int s = 0; s < 16; s++) x[s] = gg_buf_ld(rx, lanepart, (unsigned)(((s & 3) + 8 * (s >> 2)) * ldx) * 4u);
        };
        // ---- phase 1: dZ of row tile tw, channel half hw -> partial dX over those 64 channels (all four column
        //      tiles), and into the transposed LDS tile ----
        ggm_f32x16 accx[4];
        ggm_zero<4>(accx);
        auto mmq = [&](const Quad &q_, int qi) {
            const int k0 = kw + (qi >> 2) * 32 + (qi & 3) * 4;
            const gg_f32x4 a = gg_dz4v(q_.z, q_.g, 0u, 0, false, cst, C, k0);
            float *tp = Tw + k0 * GG_BF_TS + l31;
            tp[0] = a.x; tp[GG_BF_TS] = a.y; tp[2 * GG_BF_TS] = a.z; tp[3 * GG_BF_TS] = a.w;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int s = hw * 32 + qi * 4 + i;
                const gg_f32x4 b = gg_ld_f4(Wl + (size_t)(s * 64 + lane) * 4);
                accx[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(gg_bf_f4(a, i), b.x, accx[0], 0, 0, 0);
                accx[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(gg_bf_f4(a, i), b.y, accx[1], 0, 0, 0);
                accx[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(gg_bf_f4(a, i), b.z, accx[2], 0, 0, 0);
                accx[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(gg_bf_f4(a, i), b.w, accx[3], 0, 0, 0);
            }
        };
        float xb[2][16];
#pragma unroll
        for (int qi = 0; qi < 8; qi++) {
            __builtin_amdgcn_sched_barrier(0);
            mmq(Q[qi & 3], qi);
            __builtin_amdgcn_sched_barrier(0);
            if (qi < 4) ldq(Q[qi & 3], r0, qi + 4);
            else if (qi == 4) ldx16(xb[0], b0, lcb);                       // (phase 2's tiles, into registers the
            else if (qi == 6) ldx16(xb[1], b0 + 32, lcb);                  //  ring no longer needs)
        }
        __builtin_amdgcn_sched_barrier(0);
        half_barrier();
        // ---- phase 2: dW[all 128 channels, column tile ct] += dZ^T act(X) over row tiles 2hw, 2hw + 1: one formed
        //      act(X) value feeds four MFMAs (the four channel slices) ----
        float xo[2][16];
        const float bps = pcs[ct * 32 + l31], bpsh = pcs[128 + ct * 32 + l31];
#pragma unroll
        for (int jj = 0; jj < 2; jj++) {
            const int j = jj;
            float xa[16];
#pragma unroll
            for (int s = 0; s < 16; s++) xa[s] = fmaxf(__builtin_fmaf(xb[jj][s], bps, bpsh), lo);
            // (the raw X rows the epilogue needs -- row tile tw, my two column tiles: the first rides behind the second
            //  tile's MFMAs, the second is requested with the next block's chunks)
            // (the raw X rows the epilogue needs -- row tile tw, my two column tiles -- requested where the tile's own
            //  X registers have just been turned into the operand)
            ldx16(xo[jj], r0, lcd[jj]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cs = 0; cs < 4; cs++) {
                gg_f32x4 t4[4];
                const float *tp = Th + ((size_t)j * C + 32 * cs + l31) * GG_BF_TS + 4 * h;
#pragma unroll
                for (int jg = 0; jg < 4; jg++) t4[jg] = gg_ld_f4(tp + 8 * jg);
#pragma unroll
                for (int s = 0; s < 16; s++)
                    accw[cs] = __builtin_amdgcn_mfma_f32_32x32x2f32(gg_bf_f4(t4[s >> 2], s & 3), xa[s], accw[cs], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        half_barrier();
        // ---- dX exchange: the partner (same row tile, other channel half) gets my partial of ITS two column tiles ----
        // (two copies of the code under a wave-uniform branch: with `hw ? a : b` on the accumulator arrays the compiler
        //  selects register by register into copies, 64 more live registers)
        auto xsend = [&](const ggm_f32x16 &s0, const ggm_f32x16 &s1) {
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                const gg_f32x4 v0 = {s0[r], s0[r + 1], s0[r + 2], s0[r + 3]};
                const gg_f32x4 v1 = {s1[r], s1[r + 1], s1[r + 2], s1[r + 3]};
                *(gg_f32x4 *)(Xw + ((r >> 2) * 64 + lane) * 4) = v0;
                *(gg_f32x4 *)(Xw + ((4 + (r >> 2)) * 64 + lane) * 4) = v1;
            }
        };
        if (hw == 0) xsend(accx[2], accx[3]); else xsend(accx[0], accx[1]);
        long long nrow0;
        {
            // the next block's two chunks (past the end: this block's again, a harmless re-read) -- requested here, where
            // the half of the dX accumulators that went to the partner is dead: in phase 2 they did not fit
            const long long nb = blk + bstep < nblk ? blk + bstep : blk;
            nrow0 = (nb << 6) + tw * 32;
#pragma unroll
            for (int qi = 0; qi < 4; qi++) ldq(Q[qi], nrow0, qi);
        }
        half_barrier();
        // ---- epilogue: dX of row tile tw, column tiles 2hw, 2hw+1 (channel halves added in the order 0, 1) + the
        //      BatchNorm-backward sums of the layer in front (xo = its raw output there) ----
        const gg_rsrc xs = gg_make_rsrc(p.dX + (r0 * ldx + col0));
        auto finish = [&](const ggm_f32x16 &mine, int t, bool first) {
            float dxv[16];
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                const gg_f32x4 o = gg_ld_f4(Xp + ((t * 4 + (r >> 2)) * 64 + lane) * 4);
                // (k < 64 first, then k >= 64, whichever wave finishes the tile: a + b is commutative, bit for bit)
                dxv[r] = mine[r] + o.x; dxv[r + 1] = mine[r + 1] + o.y;
                dxv[r + 2] = mine[r + 2] + o.z; dxv[r + 3] = mine[r + 3] + o.w;
            }
            if (tbn[t]) {
                gg_f32x2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
                const int cc = (2 * hw + t) * 32 + l31;
                const float ps_ = pcs[cc], psh_ = pcs[128 + cc], pr_ = pcs[384 + cc], pc_ = -(pcs[256 + cc] * pr_);
                const gg_f32x2 ps2 = {ps_, ps_}, psh2 = {psh_, psh_}, pr2 = {pr_, pr_};
                const gg_f32x2 pc2 = {pc_, pc_};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const gg_f32x2 dx = {dxv[r], dxv[r + 1]}, zp = {xo[t][r], xo[t][r + 1]};
                    gg_buf_st(dx.x, xs, lcd[t], (unsigned)(((r & 3) + 8 * (r >> 2)) * ldx) * 4u);
                    gg_buf_st(dx.y, xs, lcd[t], (unsigned)((((r + 1) & 3) + 8 * ((r + 1) >> 2)) * ldx) * 4u);
                    const gg_f32x2 y = __builtin_elementwise_fma(zp, ps2, psh2);
                    const gg_f32x2 d = {y.x > 0.f ? dx.x : 0.f, y.y > 0.f ? dx.y : 0.f};
                    s1 += d;
                    s2 = __builtin_elementwise_fma(d, __builtin_elementwise_fma(zp, pr2, pc2), s2);
                }
                a1[t] += s1.x + s1.y;
                a2[t] += s2.x + s2.y;
            } else {
#pragma unroll
                for (int r = 0; r < 16; r++)
                    gg_buf_st(dxv[r], xs, lcd[t], (unsigned)(((r & 3) + 8 * (r >> 2)) * ldx) * 4u);
            }
        };
        if (hw == 0) { finish(accx[0], 0, true); finish(accx[1], 1, true); }
        else { finish(accx[2], 0, false); finish(accx[3], 1, false); }
        half_barrier();                // (the exchange area is the next block's transposed tile)
    }
    // dW partial of this workgroup: [wave][channel slice cs][reg][lane]; wave = 4 * half + column tile ct
    {
        float *out = part + ((size_t)blockIdx.x * 8 + wave) * 4096 + lane;
#pragma unroll
        for (int cs = 0; cs < 4; cs++)
#pragma unroll
            for (int r = 0; r < 16; r++) out[(cs * 16 + r) * 64] = accw[cs][r];
    }
    if (!prevbn) return;
    __syncthreads();
    float *red = lds;                                  // [8 waves][2][64]
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const float s1 = a1[t] + __shfl_xor(a1[t], 32, 64);
        const float s2 = a2[t] + __shfl_xor(a2[t], 32, 64);
        if (lane < 32) {
            red[(wave * 2 + 0) * 64 + t * 32 + lane] = s1;
            red[(wave * 2 + 1) * 64 + t * 32 + lane] = s2;
        }
    }
    __syncthreads();
    if (tid < 256) {
        const int which = tid >> 7, c = tid & 127;
        const int ct = c >> 5, hq = ct >> 1, t = ct & 1, l = c & 31;
        const int lim = p.nbn ? p.nbn : p.ndx;
        if (col0 + c < lim) {
            float v = 0.f;
            for (int w = 0; w < 4; w++)      // the four waves that finish column half hq: (half, row tile) = (w >> 1, w & 1)
                v += red[(((w >> 1) * 4 + (w & 1) + 2 * hq) * 2 + which) * 64 + t * 32 + l];
            atomicAdd(&p.psums[which * (p.nbn ? p.nbn : p.cin) + col0 + c], (double)v);
        }
    }
}

// dW[ch][col0 + col] = sum over the workgroups of their partial element, in a fixed order (bit-reproducible): one
// 1024-thread workgroup per (wave's tile, register); 16 groups of 64 lanes walk the workgroups with eight loads in
// flight, LDS adds the groups in order.  Block 0 also writes the layer's BatchNorm-backward vectors (GGLinBwd.bsums).
__global__ __launch_bounds__(1024) void gg_k_bwd_fused128_reduce(const float *__restrict__ part, int nwg,
                                                                 float *__restrict__ dW, int cin_w, int col0,
                                                                 const double *__restrict__ bsums, long long E,
                                                                 float *__restrict__ fm1, float *__restrict__ fm2,
                                                                 float *__restrict__ fdg, float *__restrict__ fdb)
{
    __shared__ float sh[16][64];
    if (bsums && blockIdx.x == 0 && threadIdx.x < GG_BF_C)
        gg_bn_bwd_fin_write(bsums, E, GG_BF_C, threadIdx.x, fm1, fm2, fdg, fdb);
    // block = (column tile ct, channel slice cs, register r); a workgroup holds TWO partials of it (row groups 0, 1)
    const int ct = blockIdx.x >> 6, cs = (blockIdx.x >> 4) & 3, r = blockIdx.x & 15;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const size_t eo = (size_t)ct * 4096 + (size_t)cs * 1024 + r * 64 + lane;     // wave = ct (+ 4: row group 1)
    const int n2 = 2 * nwg;                                                       // item i: workgroup i >> 1, row group i & 1
    float v = 0.f;
    int i = grp;
    for (; i + 7 * 16 < n2; i += 8 * 16) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = i + 16 * u;
            t[u] = part[(size_t)(k >> 1) * 32768 + (size_t)(k & 1) * 16384 + eo];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) v += t[u];
    }
    for (; i < n2; i += 16) v += part[(size_t)(i >> 1) * 32768 + (size_t)(i & 1) * 16384 + eo];
    sh[grp][lane] = v;
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
        for (int g = 0; g < 16; g++) t += sh[g][lane];
        const int ch = 32 * cs + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = col0 + 32 * ct + (lane & 31);
        dW[(size_t)ch * cin_w + col] = t;
    }
}

static int gg_bwd_fused128_grid(long long E)
{
    const long long nblk = (E >> 6) / 2;         // (two 64-row blocks per workgroup and round: one per half)
    return (int)(nblk < 256 ? (nblk < 1 ? 1 : nblk) : 256);       // one 8-wave workgroup per CU (LDS: 140 KB)
}

bool gg_linear_bwd_fused128_ok(const GGLinBwd &p)
{
    return g_bwd_fused128 && p.dX && p.Wdx && !p.amax && p.dY && p.C == GG_BF_C &&
           (p.cin == 128 || (p.cin == 256 && g_bwd_fused128 == 1)) &&
           p.ndx == p.cin && p.cin_w == p.cin && p.rot == 0 && (p.E & 127) == 0 && p.E >= 32768 && !p.zfmt &&
           !p.drop_thr && !gg_get_mlp_bf16() && p.dx_col0 == 0 && p.dx_wstride == 1 && (p.nbn == 0 || p.nbn == 128) &&
           p.dWpart && p.dW && (!p.pscale || p.psums);
}

size_t gg_linear_bwd_fused128_workspace(long long E)
{
    return (size_t)gg_bwd_fused128_grid(E) * 32768 * sizeof(float);
}

// 0 = done (dX, dW, psums, the BatchNorm-backward vectors), 1 = not this kernel's shape
int gg_linear_bwd_fused128(const GGLinBwd &pin, hipStream_t st)
{
    if (!gg_linear_bwd_fused128_ok(pin)) return 1;
    static GGDevOnce attr_done;
    const size_t lds = (size_t)(64 * 64 * 4 + 5 * GG_BF_C + 4 * 128 + 4 * GG_BF_C * GG_BF_TS + 4) * sizeof(float);
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)gg_k_linear_bwd_fused128, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) return 3;
        attr_done = true;
    }
    const int nwg = gg_bwd_fused128_grid(pin.E);
    for (int c0 = 0; c0 < pin.cin; c0 += 128) {
        GGLinBwd p = pin;
        p.dx_col0 = c0;
        p.dx_wstride = pin.cin == 256 ? 2 : 1;
        gg_k_linear_bwd_fused128<<<nwg, 512, lds, st>>>(p, p.dWpart);
        gg_k_bwd_fused128_reduce<<<256, 1024, 0, st>>>(p.dWpart, nwg, p.dW, p.cin_w, c0, c0 == 0 ? p.bsums : nullptr,
                                                       p.E, p.fin_m1, p.fin_m2, p.fin_dgamma, p.fin_dbeta);
    }
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
