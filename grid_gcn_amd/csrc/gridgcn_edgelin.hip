// gridgcn_edgelin.hip -- the first conv of a GridConv edge MLP without the gathered tensor (gfx950).
//
// sub_g_update feeds concat(geo_vec, gathered neighbour features) [B, 3+Cf, O, P] to the first 1x1
// conv of its point MLP (segmentation/models/gcn_module_g_att.py:190-194, 242-250, 135).  A 1x1 conv
// is linear and the gather only copies rows, so
//     Z0[e, :] = W[:, 3:] * feat[src(e)] + W[:, :3] * geo_vec(e) + b
//              = Ysrc[src(e), :]         + Wg * geo_vec(e)       + b,     Ysrc = feat * Wf^T
// with Ysrc computed ONCE PER SOURCE POINT ([B*Nsrc, C0] -- 8 k rows at cfg4 up2) instead of once per
// edge (3.3 M rows): the E x (3+Cf) x C0 GEMM, its gathered input tensor and, in backward, the dX and
// dW GEMMs over the edges disappear.  What is left per edge is bandwidth:
//   gg_k_edge_lin0_fwd   Z0[e] = Ysrc[src(e)] + geo terms + b (one 512-byte row read from L2, one
//                        written), att_vec rows, sum z / sum z^2 of the layer's BatchNorm
//   gg_k_edge_lin0_bwd   dZ0 formed on the fly (BN/ReLU backward of the upstream gradient, dense or
//                        the sparse arg-max form) and summed per source row over the SORTED edges
//                        (gridgcn_csr.h) -> dYsrc[B*Nsrc, C0]; dWg = sum_e geo_vec(e)^T dZ0[e]
// The two small GEMMs on [B*Nsrc] rows (Ysrc; dfeat = dYsrc * Wf, dWf = dYsrc^T * feat) run on the
// ordinary linear kernels.  Same fp32 math as the conv on the gathered tensor up to summation order.
#include "gridgcn_csr.h"
#include "gridgcn_once.h"
#include "gridgcn_edgelin.h"
#include "gridgcn_fixpt.h"

template <int VPL> struct GGV;
template <> struct GGV<1> { typedef float T; };
template <> struct GGV<2> { typedef float2 T; };
template <> struct GGV<4> { typedef float4 T; };

template <int VPL>
__global__ __launch_bounds__(256) void gg_k_edge_lin0_fwd(GGEdgeLin0 p, int epw)
{
    __shared__ float red[4][2][256];
    typedef typename GGV<VPL>::T V;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    const int c = lane * VPL, C0 = p.C0;
    const bool live = c < C0;
    const int cl = live ? c : 0;
    float w0[VPL], w1[VPL], w2[VPL], bb[VPL], s[VPL], q[VPL];
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        w0[i] = (p.Wg && live) ? p.Wg[cl + i] : 0.f;
        w1[i] = (p.Wg && live) ? p.Wg[C0 + cl + i] : 0.f;
        w2[i] = (p.Wg && live) ? p.Wg[2 * C0 + cl + i] : 0.f;
        bb[i] = live ? p.b[cl + i] : 0.f;
        s[i] = 0.f; q[i] = 0.f;
    }
    const long long rows = (long long)p.B * p.Nsrc;
    const int e0 = wid * epw, e1 = (e0 + epw < p.E) ? e0 + epw : p.E;
    // 64 edges at a time: every lane resolves ONE edge (index, source row, centre, geo_vec, its
    // att_vec row) with ordinary vector loads, then the wave walks the 64 edges with the row
    // addresses broadcast from those lanes -- no dependent scalar-load chain per edge, 8 row loads
    // in flight
    constexpr int G = 8;
    for (int eb = e0; eb < e1; eb += 64) {
        const int nloc = e1 - eb < 64 ? e1 - eb : 64;
        const bool ok = lane < nloc;
        const int ee = ok ? eb + lane : e1 - 1;
        const int ci = ee / p.P, bi = ci / p.O;
        long long flat = (long long)p.nebidx[ee] + (long long)bi * p.Nsrc;
        flat = flat < 0 ? 0 : (flat > rows - 1 ? rows - 1 : flat);
        const float *srow = p.src + flat * p.Cs;
        const float *cen = p.cent + (size_t)ci * p.cent_stride;
        const float nx = srow[0], ny = srow[1], nz = srow[2];
        const float cx = cen[0], cy = cen[1], cz = cen[2];
        const float gx = nx - cx, gy = ny - cy, gz = nz - cz;
        if (ok) {
            float4 *a = (float4 *)(p.att16 + (size_t)ee * 16);
            a[0] = make_float4(sqrtf((gx * gx + gy * gy) + gz * gz), gx, gy, gz);
            a[1] = make_float4(cx, cy, cz, nx);
            a[2] = make_float4(ny, nz, 0.f, 0.f);
            a[3] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int flat_i = (int)flat;                      // B*Nsrc < 2^31 (checked by the index ops)
        if (!p.Z && !p.sums) continue;                     // evaluation: only att16 is wanted
        for (int j = 0; j < nloc; j += G) {
            V y[G];
            float gxj[G], gyj[G], gzj[G];
#pragma unroll
            for (int u = 0; u < G; u++) {
                const int jj = j + u < nloc ? j + u : nloc - 1;
                const int fl = __builtin_amdgcn_readlane(flat_i, jj);
                gxj[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gx), jj));
                gyj[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gy), jj));
                gzj[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gz), jj));
                if (p.Ysrc) y[u] = *(const V *)(p.Ysrc + (size_t)fl * C0 + cl);
            }
#pragma unroll
            for (int u = 0; u < G; u++) {
                if (j + u >= nloc) break;
                float z[VPL];
                const float *yf = (const float *)&y[u];
#pragma unroll
                for (int i = 0; i < VPL; i++) {
                    float v = p.Ysrc ? yf[i] : 0.f;
                    v = fmaf(gxj[u], w0[i], v);
                    v = fmaf(gyj[u], w1[i], v);
                    v = fmaf(gzj[u], w2[i], v);
                    v += bb[i];
                    z[i] = v;
                    s[i] += v;
                    q[i] += v * v;
                }
                if (live && p.Z) *(V *)(p.Z + (size_t)(eb + j + u) * C0 + c) = *(const V *)z;
            }
        }
    }
    // batch statistics: lanes -> waves (LDS) -> fp64 atomics
    if (!p.sums) return;
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        red[wave][0][lane * VPL + i] = live ? s[i] : 0.f;
        red[wave][1][lane * VPL + i] = live ? q[i] : 0.f;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * C0; t += 256) {
        const int which = t / C0, col = t - which * C0;
        const float v = (red[0][which][col] + red[1][which][col]) + (red[2][which][col] + red[3][which][col]);
        atomicAdd(&p.sums[which * C0 + col], (double)v);
    }
}

template <int VPL>
__global__ __launch_bounds__(256) void gg_k_edge_lin0_bwd(GGEdgeLin0Bwd p)
{
    __shared__ float red[4][3][256];
    typedef typename GGV<VPL>::T V;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    const int C0 = p.C0, N = p.N, M = p.M;
    const int c = lane * VPL;
    const bool live = c < C0;
    const int cl = live ? c : 0;
    float sc[VPL], sh[VPL], mu[VPL], bz[VPL], cz[VPL], wg[3][VPL], acc[VPL];
    float w0[VPL], w1[VPL], w2[VPL], bb[VPL];        // forward constants (Z recomputed)
    // (every constant loaded unconditionally from a valid address -- cl is clamped, an absent table reads the
    //  scale vector instead -- and selected afterwards: a load behind its own condition is a branch with a wait,
    //  nine memory round trips in a row at the start of this kernel)
    {
        const bool rw = !p.Z && live, hw = rw && p.Wg, hb = rw && p.b;
        const float *wgp = p.Wg ? p.Wg : p.scale;          // [3][C0] or, absent, anything readable at [cl + i]
        const int cs = p.Wg ? C0 : 0;
        const float *bp = p.b ? p.b : p.scale;
#pragma unroll
        for (int i = 0; i < VPL; i++) {
            const float a0 = wgp[cl + i], a1 = wgp[cs + cl + i], a2 = wgp[2 * cs + cl + i], a3 = bp[cl + i];
            const float sv = p.scale[cl + i], shv = p.shift[cl + i], muv = p.mean[cl + i], rsv = p.rstd[cl + i],
                        m1v = p.m1[cl + i], m2v = p.m2[cl + i];
            w0[i] = hw ? a0 : 0.f; w1[i] = hw ? a1 : 0.f; w2[i] = hw ? a2 : 0.f; bb[i] = hb ? a3 : 0.f;
            const float s = live ? sv : 0.f;
            sc[i] = s; sh[i] = live ? shv : 0.f; mu[i] = live ? muv : 0.f;
            bz[i] = live ? -(s * rsv) * m2v : 0.f;
            cz[i] = live ? -(s * m1v) : 0.f;
            wg[0][i] = 0.f; wg[1][i] = 0.f; wg[2][i] = 0.f; acc[i] = 0.f;
        }
    }
    const bool active = wid < p.B * p.cpc;
    if (active) {
        const int b = wid / p.cpc, ch = wid - b * p.cpc;
        const int e0 = ch * p.chunk, e1 = e0 + p.chunk < M ? e0 + p.chunk : M;
        const long long rows = (long long)p.B * N;
        const int *pk = p.keys + (size_t)b * M, *pp = p.perm + (size_t)b * M;
        const int *rp = p.rowptr + (size_t)b * (N + 3);
        const size_t ebase = (size_t)b * M;
        int cur = pk[e0], rs = e0;

        auto flush = [&](int key, int rbeg, int rend) {
            if (key > N || !live) return;
            long long dest = (long long)b * N - 1 + key;
            if (dest < 0) dest = 0;
            float *d = p.dYsrc + dest * C0 + c;
            const int r0 = rp[key], r1 = rp[key + 1];      // (both loads together: rp has N + 3 entries, key <= N)
            const bool whole = (key != 0) & (key != N) & (rbeg == r0) & (rend == r1);
            if (whole) {
#pragma unroll
                for (int i = 0; i < VPL; i++) d[i] = acc[i];
            } else {
#pragma unroll
                for (int i = 0; i < VPL; i++) atomicAdd(&d[i], acc[i]);
            }
        };

        // 64 sorted edges at a time: every lane resolves one edge (perm, key, centre, geo_vec) with
        // vector loads; the wave then walks them with the values broadcast from those lanes
        constexpr int G = 4;
        for (int eb = e0; eb < e1; eb += 64) {
        const int nloc = e1 - eb < 64 ? e1 - eb : 64;
        const int el = lane < nloc ? eb + lane : e1 - 1;
        const int k_l = pk[el], m_l = pp[el];
        const int o_l = m_l / p.P, pn_l = m_l - o_l * p.P;
        const float4 a_l = *(const float4 *)(p.att16 + (ebase + m_l) * 16);
        for (int jb = 0; jb < nloc; jb += G) {
            const int e = eb + jb;
            int k[G], m[G], pn[G];
            V z[G], g[G];
            int am[G][VPL];
            float gx[G], gy[G], gz[G];
#pragma unroll
            for (int j = 0; j < G; j++) {
                const int jj = jb + j < nloc ? jb + j : nloc - 1;
                k[j] = __builtin_amdgcn_readlane(k_l, jj);
                m[j] = __builtin_amdgcn_readlane(m_l, jj);
                pn[j] = __builtin_amdgcn_readlane(pn_l, jj);
                gx[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a_l.y), jj));
                gy[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a_l.z), jj));
                gz[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a_l.w), jj));
                const size_t row = ebase + m[j];
                if (p.Z) {
                    z[j] = *(const V *)(p.Z + row * C0 + cl);
                } else {
                    // recompute as the forward did: source row = destination row of the edge
                    long long flat;
                    if (k[j] <= N) {
                        flat = (long long)b * N - 1 + k[j];
                        if (flat < 0) flat = 0;
                    } else {
                        flat = (long long)p.index[row] + (long long)b * N;
                        flat = flat < 0 ? 0 : (flat > rows - 1 ? rows - 1 : flat);
                    }
                    float zr[VPL];
                    if (p.Ysrc) {
                        const V y = *(const V *)(p.Ysrc + flat * C0 + cl);
                        const float *yf = (const float *)&y;
#pragma unroll
                        for (int i = 0; i < VPL; i++) zr[i] = yf[i];
                    } else {
#pragma unroll
                        for (int i = 0; i < VPL; i++) zr[i] = 0.f;
                    }
#pragma unroll
                    for (int i = 0; i < VPL; i++) {
                        float v = zr[i];
                        v = fmaf(gx[j], w0[i], v);
                        v = fmaf(gy[j], w1[i], v);
                        v = fmaf(gz[j], w2[i], v);
                        zr[i] = v + bb[i];
                    }
                    z[j] = *(const V *)zr;
                }
                if (p.dY) {
                    g[j] = *(const V *)(p.dY + row * C0 + cl);
                } else {
                    const int o = __builtin_amdgcn_readlane(o_l, jj);
                    const size_t ar = ((size_t)b * p.O + o) * C0 + cl;
                    g[j] = *(const V *)(p.gval + ar);
#pragma unroll
                    for (int i = 0; i < VPL; i++) am[j][i] = p.amax[ar + i];
                }
            }
#pragma unroll
            for (int j = 0; j < G; j++) {
                if (e + j >= e1) break;
                const float *zf = (const float *)&z[j], *gf = (const float *)&g[j];
                float dz[VPL];
#pragma unroll
                for (int i = 0; i < VPL; i++) {
                    float gg = gf[i];
                    if (!p.dY) gg = am[j][i] == pn[j] ? gg : 0.f;
                    dz[i] = sc[i] * ((zf[i] * sc[i] + sh[i] > 0.f) ? gg : 0.f) +
                            ((zf[i] - mu[i]) * bz[i] + cz[i]);
                    wg[0][i] = fmaf(gx[j], dz[i], wg[0][i]);
                    wg[1][i] = fmaf(gy[j], dz[i], wg[1][i]);
                    wg[2][i] = fmaf(gz[j], dz[i], wg[2][i]);
                }
                if (k[j] == N + 1) {
                    if (cur != N + 1) { flush(cur, rs, e + j); cur = N + 1; }
                    if (live) {
                        long long flat = (long long)p.index[ebase + m[j]] + (long long)b * N;
                        flat = flat < 0 ? 0 : (flat > rows - 1 ? rows - 1 : flat);
#pragma unroll
                        for (int i = 0; i < VPL; i++) atomicAdd(&p.dYsrc[flat * C0 + c + i], dz[i]);
                    }
                    continue;
                }
                if (k[j] != cur) {
                    flush(cur, rs, e + j);
                    cur = k[j];
                    rs = e + j;
#pragma unroll
                    for (int i = 0; i < VPL; i++) acc[i] = 0.f;
                }
#pragma unroll
                for (int i = 0; i < VPL; i++) acc[i] += dz[i];
            }
        }
        }
        flush(cur, rs, e1);
    }
    if (!p.dWg) return;
#pragma unroll
    for (int i = 0; i < VPL; i++)
#pragma unroll
        for (int t = 0; t < 3; t++) red[wave][t][lane * VPL + i] = live ? wg[t][i] : 0.f;
    __syncthreads();
    for (int t = threadIdx.x; t < 3 * C0; t += 256) {
        const int which = t / C0, col = t - which * C0;
        const float v = (red[0][which][col] + red[1][which][col]) + (red[2][which][col] + red[3][which][col]);
        atomicAdd(&p.dWg[which * C0 + col], (double)v);
    }
}

static int vpl_for(int C0) { return C0 <= 64 ? 1 : (C0 <= 128 ? 2 : 4); }

// ------------------------------------------------------------------------------------------
// Batch statistics of a single-layer point MLP WITHOUT the edge x channel pass.  z0[e, c] =
// y[n(e), c] + w_c . g(e) (y = Ysrc + b per source point, g = geo_vec, w_c = the layer's three geo
// weights) is affine in per-source and per-edge quantities, so with cnt(n) = #edges of source n,
// G(n) = sum of their geo_vec and GG = sum_e g g^T
//   sum_e z0   = sum_n cnt(n) y[n,c] + w_c . G(n)
//   sum_e z0^2 = sum_n (cnt(n) y[n,c]^2 + 2 y[n,c] w_c . G(n)) + w_c^T GG w_c
// -- 8 K source rows x C instead of 3.3 M edges x C at cfg4 up2 (gg_k_edge_lin0_fwd: 186 us, VALU bound
// on 6 flops per edge and channel).  What is left per edge is what the attention branch needs anyway:
//   gg_k_edge_geo_fwd    grid (nsplit, B): the att_vec row of every edge (as gg_k_edge_lin0_fwd writes
//                        it) and, in LDS, cnt / G per source of the cloud -- 64-bit fixed-point integer
//                        atomics exactly as the geo pass of gg_k_edge_lin0_bwd_sparse (scale from the
//                        first round, fp32 global side path beyond the headroom or for an index clipped
//                        into another cloud) -> gpart[B][nsplit][N+1][4]; GG and sum g by fp64 atomics
//   gg_k_edge_geo_stats  per block of 32 source rows: Gsum[r] = (G, cnt) summed over the splits in a
//                        fixed order, then the two sums above per channel (fp64 atomics into `sums`)
struct GGEdgeGeoFwd {
    const float *src;     // [B*N][Cs]
    const int *nebidx;    // [B][O*P]
    const float *cent;    // centre ci at cent + ci*cent_stride
    float *att16;         // [E][16]
    float *gpart;         // [B][nsplit][N+1][4]  (gx, gy, gz, count)
    float *fgs;           // [B*N][4], zero-filled
    double *gg;           // [12] += (sum g_j g_k [9], sum g_j [3]), zero-filled
    int cent_stride, B, N, Cs, O, P, nsplit, epw;   // epw edges per workgroup (a multiple of 1024)
};

__global__ __launch_bounds__(1024, 8) void gg_k_edge_geo_fwd(GGEdgeGeoFwd p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float red[16][12];
    __shared__ unsigned gmax;
    const int N = p.N, O = p.O, P = p.P;
    const int sp = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    long long *gx = (long long *)lds, *gy = gx + (N + 1), *gz = gy + (N + 1);
    int *gc = (int *)(gz + (N + 1));
    const long long rows = (long long)p.B * N;
    // the cloud's O*P edges in runs of epw per workgroup.  FOUR lanes per edge: lane q of the quad writes
    // the q-th 16-byte piece of the edge's att_vec row (a wave's store is 1 KB of consecutive bytes; one
    // lane per edge wrote four pieces 64 bytes apart, 256 L1 transactions per 64 edges instead of 64) and
    // adds the q-th of (gx, gy, gz, count); the quad's loads hit the same addresses
    const int ec = O * P;
    const int ea = sp * p.epw < ec ? sp * p.epw : ec, ez = ea + p.epw < ec ? ea + p.epw : ec;
    const int *nb = p.nebidx + (size_t)b * O * P;
    float *ab = p.att16 + (size_t)b * O * P * 16;
    const int q = tid & 3, et = tid >> 2;                          // 256 edges per round of the workgroup
    for (int i = tid; i <= N; i += 1024) { gx[i] = 0; gy[i] = 0; gz[i] = 0; gc[i] = 0; }
    if (tid == 0) gmax = 0u;
    const bool v4 = (p.cent_stride & 3) == 0 && (p.Cs & 3) == 0 && (((size_t)p.cent | (size_t)p.src) & 15) == 0;
    // one edge: source row (mx.sym.take clip mode, utils/ops.py:78-93), centre.  32-bit arithmetic throughout
    // (B * N and B * O * P are below 2^31: checked by the entry): the 64-bit forms were a third of the kernel's
    // instructions.  ci = e / P is kept incrementally by the caller.
    const int rowsm1 = (int)rows - 1, bN = b * N;
    auto edge = [&](int e, int ci, int &flat, float4 &s4, float4 &c4) {
        int f = nb[e] + bN;
        f = f < 0 ? 0 : (f > rowsm1 ? rowsm1 : f);
        flat = f;
        const float *srow = p.src + (size_t)f * p.Cs;
        const float *cen = p.cent + ((size_t)b * O + ci) * p.cent_stride;
        if (v4) {
            s4 = *(const float4 *)srow;
            c4 = *(const float4 *)cen;
        } else {
            s4 = make_float4(srow[0], srow[1], srow[2], 0.f);
            c4 = make_float4(cen[0], cen[1], cen[2], 0.f);
        }
    };
    auto geo = [&](const float4 s4, const float4 c4) -> float4 {
        const float x = s4.x - c4.x, y = s4.y - c4.y, z = s4.z - c4.z;
        return make_float4(sqrtf((x * x + y * y) + z * z), x, y, z);
    };
    // scale of the three geo sums from the first round of edges (count: exact integers)
    float gm = 0.f;
    if (ea + et < ez) {
        int fl;
        float4 s4, c4;
        edge(ea + et, (ea + et) / P, fl, s4, c4);
        const float4 g = geo(s4, c4);
        gm = fmaxf(fmaxf(fabsf(g.y), fabsf(g.z)), fabsf(g.w));
        if (!(gm < 3.0e38f)) gm = 0.f;                            // (NaN / inf never set the scale)
    }
    __syncthreads();
    atomicMax(&gmax, __float_as_uint(gm));
    __syncthreads();
    const int gk = gg_fix_exp(gmax);
    const float gF = ldexpf(1.f, gk);
    const float gthr = fminf(0x1p+47f, 0x1p+62f / ((float)p.epw + 1.f));
    float g4[4] = {0.f, 0.f, 0.f, 0.f};      // lane q < 3: sum g_q g_0, g_q g_1, g_q g_2, g_q
    constexpr int UG = 4;
    // centre of edge e + 256 u, kept without a division per edge: (quotient, remainder) of 256 by P once
    const int q256 = 256 / P, r256 = 256 - q256 * P;
    int ci0 = (ea + et) / P, rm0 = (ea + et) - ci0 * P;
    const int keyoff = bN - 1;
    for (int e = ea + et; e < ez; e += 256 * UG) {
        int fl[UG], ci[UG];
        float4 s4[UG], c4[UG];
#pragma unroll
        for (int u = 0; u < UG; u++) {
            ci[u] = ci0;
            ci0 += q256; rm0 += r256;
            if (rm0 >= P) { rm0 -= P; ci0++; }
        }
#pragma unroll
        for (int u = 0; u < UG; u++) {
            const bool in = e + 256 * u < ez;
            edge(in ? e + 256 * u : e, in ? ci[u] : ci[0], fl[u], s4[u], c4[u]);
        }
#pragma unroll
        for (int u = 0; u < UG; u++) {
            if (e + 256 * u >= ez) break;
            const float4 g = geo(s4[u], c4[u]);
            const float4 piece = q == 0 ? g : (q == 1 ? make_float4(c4[u].x, c4[u].y, c4[u].z, s4[u].x)
                                                      : (q == 2 ? make_float4(s4[u].y, s4[u].z, 0.f, 0.f)
                                                                : make_float4(0.f, 0.f, 0.f, 0.f)));
            *(float4 *)(ab + (size_t)(e + 256 * u) * 16 + 4 * q) = piece;
            // key in [0, N] = destination row - (b*N - 1); a row of ANOTHER cloud (never produced by the
            // index ops) goes to the global side buffer by its flat row
            const int li = fl[u] - keyoff;
            const int key = (li < 0 || li > N) ? -1 : li;
            const float mine = q == 0 ? g.y : (q == 1 ? g.z : g.w);                       // (q < 3)
            const float x0 = g.y * gF, x1 = g.z * gF, x2 = g.w * gF;
            const bool fits = fabsf(x0) < gthr && fabsf(x1) < gthr && fabsf(x2) < gthr;   // (false for NaN)
            if (key >= 0 && fits) {
                if (q == 3) atomicAdd(&gc[key], 1);
                else
                    atomicAdd((unsigned long long *)(q == 0 ? &gx[key] : (q == 1 ? &gy[key] : &gz[key])),
                              (unsigned long long)gg_fix_i64(mine * gF));
            } else {
                atomicAdd(&p.fgs[(size_t)fl[u] * 4 + q], q == 3 ? 1.f : mine);
            }
            if (q < 3) { g4[0] += mine * g.y; g4[1] += mine * g.z; g4[2] += mine * g.w; g4[3] += mine; }
        }
    }
    __syncthreads();
    float *gp_ = p.gpart + (((size_t)b * p.nsplit + sp) * (N + 1)) * 4;
    const float gi = ldexpf(1.f, -gk);
    for (int i = tid; i <= N; i += 1024)
        ((float4 *)gp_)[i] = make_float4((float)gx[i] * gi, (float)gy[i] * gi, (float)gz[i] * gi, (float)gc[i]);
    // GG / sum g: lanes of equal q -> waves -> one fp64 atomic per value and workgroup
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float v = q < 3 ? g4[i] : 0.f;
#pragma unroll
        for (int o = 32; o >= 4; o >>= 1) v += __shfl_xor(v, o, 64);
        // lane q of the wave now holds the sum over the wave's lanes == q (mod 4)
        if ((tid & 63) < 3) red[tid >> 6][i < 3 ? 3 * (tid & 63) + i : 9 + (tid & 63)] = v;
    }
    __syncthreads();
    if (tid < 12) {
        float v = 0.f;
        for (int w = 0; w < 16; w++) v += red[w][tid];
        atomicAdd(&p.gg[tid], (double)v);
    }
}

#define GG_GS_ROWS 32
__global__ __launch_bounds__(256) void gg_k_edge_geo_stats(
    const float *__restrict__ gpart, const float *__restrict__ fgs, const float *__restrict__ Ysrc,
    const float *__restrict__ Wg, const float *__restrict__ bias, int B, int N, int C, int nsplit,
    float *__restrict__ Gsum, const double *__restrict__ gg, double *__restrict__ sums)
{
    __shared__ float4 sg[GG_GS_ROWS], sp8[8][GG_GS_ROWS];
    const long long rows = (long long)B * N;
    const long long r0 = (long long)blockIdx.x * GG_GS_ROWS;
    const int nr = rows - r0 < GG_GS_ROWS ? (int)(rows - r0) : GG_GS_ROWS;
    {
        // row r collects key n+1 of its own cloud and, for its last row, key 0 of the next (as
        // gg_k_edge_lin0_bwd_finish); eight thread groups take every eighth split, added up in a fixed order
        const int ri = threadIdx.x & (GG_GS_ROWS - 1), part = threadIdx.x / GG_GS_ROWS;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ri < nr) {
            const long long r = r0 + ri;
            const int b = (int)(r / N), n = (int)(r - (long long)b * N);
            for (int sp = part; sp < nsplit; sp += 8) {
                const float4 g = *(const float4 *)(gpart + (((size_t)b * nsplit + sp) * (N + 1) + (n + 1)) * 4);
                a.x += g.x; a.y += g.y; a.z += g.z; a.w += g.w;
                if (n == N - 1 && b + 1 < B) {
                    const float4 h = *(const float4 *)(gpart + ((size_t)(b + 1) * nsplit + sp) * (N + 1) * 4);
                    a.x += h.x; a.y += h.y; a.z += h.z; a.w += h.w;
                }
            }
        }
        sp8[part][ri] = a;
    }
    __syncthreads();
    if ((int)threadIdx.x < nr) {
        const long long r = r0 + threadIdx.x;
        float4 a = *(const float4 *)(fgs + r * 4);
        for (int k = 0; k < 8; k++) {
            const float4 g = sp8[k][threadIdx.x];
            a.x += g.x; a.y += g.y; a.z += g.z; a.w += g.w;
        }
        sg[threadIdx.x] = a;
        *(float4 *)(Gsum + r * 4) = a;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const float w0 = Wg ? Wg[c] : 0.f, w1 = Wg ? Wg[C + c] : 0.f, w2 = Wg ? Wg[2 * C + c] : 0.f, bc = bias[c];
        float s = 0.f, q = 0.f;
        for (int i = 0; i < nr; i++) {
            const float4 g = sg[i];
            const float y = Ysrc[(r0 + i) * C + c] + bc;
            const float gw = (g.x * w0 + g.y * w1) + g.z * w2;
            s += g.w * y + gw;
            q += (g.w * y) * y + 2.f * (y * gw);
        }
        double qd = (double)q;
        if (blockIdx.x == 0) {
            // w^T GG w: the edges' own quadratic term, once
            const double W[3] = {(double)w0, (double)w1, (double)w2};
            for (int j = 0; j < 3; j++)
                for (int k = 0; k < 3; k++) qd += W[j] * W[k] * gg[3 * j + k];
        }
        atomicAdd(&sums[c], (double)s);
        atomicAdd(&sums[C + c], qd);
    }
}

// edges per workgroup: whole rounds of 4 x 256, at most two workgroups per CU over the batch
static int gg_edge_geo_epw(int B, long long ec)
{
    long long want = (ec * B + 511) / 512;
    want = (want + 1023) / 1024 * 1024;
    return (int)(want < 1024 ? 1024 : want);
}
static int gg_edge_geo_nsplit(int B, long long ec)
{
    const int epw = gg_edge_geo_epw(B, ec);
    return (int)((ec + epw - 1) / epw);
}

size_t gg_edge_geo_workspace(int B, int N, long long ec)
{
    return ((size_t)B * gg_edge_geo_nsplit(B, ec) * (N + 1) * 4 + (size_t)B * N * 4) * sizeof(float);
}

// 1 = shape not supported (the caller uses gg_edge_lin0_fwd).  gg[12], sums[2C] zero-filled by the caller.
int gg_edge_geo_fwd(const float *Ysrc, const float *src, const int *nebidx, const float *cent, int cent_stride,
                    int B, int N, int Cs, int O, int P, int C, const float *Wg, const float *bias, float *att16,
                    float *Gsum, double *gg, double *sums, void *workspace, hipStream_t st)
{
    const size_t lds = (size_t)(N + 1) * 28;
    if (lds > 150 * 1024 || C < 1 || C > 1024 || (long long)B * O * P >= (1ll << 31)) return 1;
    static GGDevOnce attr_done;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)gg_k_edge_geo_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess) return 3;
        attr_done = true;
    }
    GGEdgeGeoFwd p;
    p.src = src; p.nebidx = nebidx; p.cent = cent; p.att16 = att16; p.gg = gg;
    p.cent_stride = cent_stride; p.B = B; p.N = N; p.Cs = Cs; p.O = O; p.P = P;
    p.epw = gg_edge_geo_epw(B, (long long)O * P);
    p.nsplit = gg_edge_geo_nsplit(B, (long long)O * P);
    p.gpart = (float *)workspace;
    p.fgs = p.gpart + (size_t)B * p.nsplit * (N + 1) * 4;
    if (hipMemsetAsync(p.fgs, 0, (size_t)B * N * 4 * sizeof(float), st) != hipSuccess) return 3;
    gg_k_edge_geo_fwd<<<dim3(p.nsplit, B), 1024, lds, st>>>(p);
    const long long rows = (long long)B * N;
    gg_k_edge_geo_stats<<<(int)((rows + GG_GS_ROWS - 1) / GG_GS_ROWS), 256, 0, st>>>(
        p.gpart, p.fgs, Ysrc, Wg, bias, B, N, C, p.nsplit, Gsum, gg, sums);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_edge_lin0_fwd(const GGEdgeLin0 &p, hipStream_t st)
{
    if (p.C0 < 1 || p.C0 > 256 || p.E < 1) return 1;
    const int VPL = vpl_for(p.C0);
    if (p.C0 % VPL) return 1;
    // ~8 waves per SIMD worth of waves, each a contiguous range of edges
    int nwave = 256 * 4 * 8;
    if (nwave > (p.E + 31) / 32) nwave = (p.E + 31) / 32;
    if (nwave < 1) nwave = 1;
    const int epw = (p.E + nwave - 1) / nwave;
    nwave = (p.E + epw - 1) / epw;
    const int grid = (nwave + 3) / 4;
    if (VPL == 1) gg_k_edge_lin0_fwd<1><<<grid, 256, 0, st>>>(p, epw);
    else if (VPL == 2) gg_k_edge_lin0_fwd<2><<<grid, 256, 0, st>>>(p, epw);
    else gg_k_edge_lin0_fwd<4><<<grid, 256, 0, st>>>(p, epw);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int gg_edge_lin0_bwd(GGEdgeLin0Bwd p, void *workspace, hipStream_t st)
{
    if (p.C0 < 1 || p.C0 > 256) return 1;
    const int VPL = vpl_for(p.C0);
    if (p.C0 % VPL) return 1;
    int *perm, *keys, *rowptr;
    const int rc = gg_csr_build(p.index, p.B, p.N, p.M, workspace, &perm, &keys, &rowptr, st);
    if (rc) return rc;
    p.perm = perm; p.keys = keys; p.rowptr = rowptr;
    // a wave walks its chunk of sorted edges four rows at a time, each round a full memory latency:
    // short chunks where the edges are few (cfg4 down1: 64 K edges were 64 workgroups x 64 rounds),
    // rows cut by a chunk boundary go through atomics either way
    p.chunk = GG_CSR_CHUNK;
    while (p.chunk > 32 && (long long)p.B * ((p.M + p.chunk - 1) / p.chunk) < 1024) p.chunk >>= 1;
    p.cpc = (p.M + p.chunk - 1) / p.chunk;
    const int nwave = p.B * p.cpc, grid = (nwave + 3) / 4;
    if (VPL == 1) gg_k_edge_lin0_bwd<1><<<grid, 256, 0, st>>>(p);
    else if (VPL == 2) gg_k_edge_lin0_bwd<2><<<grid, 256, 0, st>>>(p);
    else gg_k_edge_lin0_bwd<4><<<grid, 256, 0, st>>>(p);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

// ------------------------------------------------------------------------------------------
// Sparse form of the same backward, for a SINGLE-layer point MLP (upstream = the max pool).
// dZ0[e,c] = [e is the arg-max edge of (o,c)] s[o,c] + (z0[e,c] - mu_c) bz_c + cz_c, and z0 is
// affine in per-source / per-edge quantities (z0 = Ysrc[src(e)] + Wg geo(e) + b), so the per-source
// sum of the DENSE part needs only cnt(n) = #edges of source n and G(n) = sum of their geo_vec:
//   dYsrc[n,c] = sum_{(o,c): src(e*) = n} s[o,c]
//              + bz_c (cnt(n) (Ysrc[n,c] + b_c) + G(n) . Wg[:,c]) + cnt(n) (cz_c - mu_c bz_c)
// The edge x channel work (420 M values at cfg4 up2) shrinks to the ncent x C arg-max entries (84 M),
// scattered with LDS atomics into a per-(cloud, 16-channel slice) copy of the destination rows.
//   gg_k_edge_lin0_bwd_sparse   grid (nsplit, C/16 + 1, B): LDS acc[(N+1)][16]; the last y row collects
//                               cnt / G per source, sum geo geo^T, sum geo; partials are stored
//   gg_k_edge_lin0_bwd_finish   dYsrc = partials + dense part; per-source (G, cnt) for dWg
// The LDS sums are 64-bit FIXED POINT: ds_add_f32 runs at ~2 cycles per active lane on gfx950 (the
// 42 M float atomics of cfg4 up2 were 270 of this kernel's 650 us, the 13 M of the geo pass another
// 230), the integer atomics ~8x faster (profiles/r3_sparse_ablation.txt).  Per channel the scale is a
// power of two set from the workgroup's first round of entries, 2^40 > max |s| * scale >= 2^39: a
// value is cut at 2^-40 of that maximum (fp32 atomics rounded every partial SUM to 2^-24 of itself),
// the integer sum is exact whatever the order -- the partials are now bit-reproducible -- and an
// entry too large for the headroom (2^47, or 2^62 / entries per workgroup if that is less) takes the
// fp32 global-atomic side path that already serves indices clipped into another cloud.
struct GGEdgeSparse {
    const int *nebidx;       // [B][O*P]
    const float *att16;      // [E][16]
    const unsigned char *amax;   // [B*O][C]
    const float *gval, *zsel;   // [B*O][C]: gradient w.r.t. relu(bn(z0)) and z0 at the arg max
    const float *sc, *sh;    // [C] BatchNorm scale / shift of the layer
    float *part;             // [B][nsplit][N+1][C]
    float *gpart;            // [B][nsplit][N+1][4]   (gx, gy, gz, count)
    float *fpart, *fgs;      // [B*N][C], [B*N][4]: zero-filled; edges clipped into another cloud
    double *wgs;             // [3][C] += sum_(o,c) geo_j(e*) s[o,c]
    double *gg;              // [12] += (sum geo_j geo_k [9], sum geo_j [3])
    int B, N, O, P, C, nsplit;
    int geo_given;           // the forward left (G, cnt) per source and gg (gg_k_edge_geo_fwd): no geo pass here
};

__global__ __launch_bounds__(1024) void gg_k_edge_lin0_bwd_sparse(GGEdgeSparse p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float red[16][16];
    __shared__ unsigned cmax[16], gmax;
    const int N = p.N, O = p.O, P = p.P, C = p.C;
    const int sp = blockIdx.x, sl = blockIdx.y, b = blockIdx.z;
    const int c0 = sl * 16, tid = threadIdx.x, lane = tid & 15, grp = tid >> 4;   // 64 groups
    long long *acc = (long long *)lds;         // [(N+1)][16]
    // the geo pass has workgroups of its own (the last y row of the grid, no channels): short ones
    // that fill gaps, instead of a tail on every eighth workgroup of the channel slices
    const bool geo_wg = !p.geo_given && sl == (int)gridDim.y - 1;
    long long *gx = (long long *)lds, *gy = gx + (N + 1), *gz = gy + (N + 1);   // [(N+1)] each: one
    int *gc = (int *)(gz + (N + 1));           // array per component (bank = key, not 4 keys per bank row)
    const long long rows = (long long)p.B * N;
    const int per = (O + p.nsplit - 1) / p.nsplit;
    const int o0 = sp * per, o1 = o0 + per < O ? o0 + per : O;
    const int c = c0 + lane;
    const bool cok = c < C;
    const float scv = cok ? p.sc[c] : 0.f, shv = cok ? p.sh[c] : 0.f;
    float wg0 = 0.f, wg1 = 0.f, wg2 = 0.f;
    float g9[12];
#pragma unroll
    for (int i = 0; i < 12; i++) g9[i] = 0.f;
    const int *nb = p.nebidx + (size_t)b * O * P;
    // key in [0, N] = destination row - (b*N - 1); an index clipped into ANOTHER cloud (never
    // produced by the index ops) goes to the global side buffers fpart / fgs by its flat row
    long long flat_ = 0;
    auto keyof = [&](int idx) -> int {
        long long flat = (long long)idx + (long long)b * N;
        flat = flat < 0 ? 0 : (flat > rows - 1 ? rows - 1 : flat);
        flat_ = flat;
        const long long li = flat - ((long long)b * N - 1);
        return (li < 0 || li > N) ? -1 : (int)li;
    };
    if (tid < 16) cmax[tid] = 0u;
    if (tid == 16) gmax = 0u;
    // per-source geo sums (channel slice 0 only): a flat walk over the edges of this split, one edge
    // per thread and four in flight (the per-centre form kept P of 32 lanes busy and made these
    // workgroups the tail of the launch)
    if (geo_wg) {                                                 // (uniform in the workgroup)
        for (int i = tid; i <= N; i += 1024) { gx[i] = 0; gy[i] = 0; gz[i] = 0; gc[i] = 0; }
        const int ea = o0 < o1 ? o0 * P : 0, ez = o0 < o1 ? o1 * P : 0;
        const float *ab = p.att16 + (size_t)b * O * P * 16;
        constexpr int UG = 4;
        // scale of the three geo sums from the first round of edges (count: exact integers)
        float gm = 0.f;
        if (ea + tid < ez) {
            const float4 a = *(const float4 *)(ab + (size_t)(ea + tid) * 16);
            gm = fmaxf(fmaxf(fabsf(a.y), fabsf(a.z)), fabsf(a.w));
            if (!(gm < 3.0e38f)) gm = 0.f;                        // (NaN / inf never set the scale)
        }
        __syncthreads();                                          // gs zeroed, gmax zeroed
        atomicMax(&gmax, __float_as_uint(gm));
        __syncthreads();
        const int gk = gg_fix_exp(gmax);
        const float gF = ldexpf(1.f, gk);
        const float gthr = fminf(0x1p+47f, 0x1p+62f / ((float)per * (float)P + 1.f));
        for (int e = ea + tid; e < ez; e += 1024 * UG) {
            float4 a[UG];
            int id[UG];
#pragma unroll
            for (int u = 0; u < UG; u++) {
                const int ee = e + 1024 * u < ez ? e + 1024 * u : e;
                a[u] = *(const float4 *)(ab + (size_t)ee * 16);
                id[u] = nb[ee];
            }
            __builtin_amdgcn_sched_barrier(0);     // (all loads out before the first one is waited for)
#pragma unroll
            for (int u = 0; u < UG; u++) {
                if (e + 1024 * u >= ez) break;
                const int key = keyof(id[u]);
                const float x0 = a[u].y * gF, x1 = a[u].z * gF, x2 = a[u].w * gF;
                const bool fits = fabsf(x0) < gthr && fabsf(x1) < gthr && fabsf(x2) < gthr;   // (false for NaN)
                if (key >= 0 && fits) {
                    atomicAdd((unsigned long long *)&gx[key], (unsigned long long)gg_fix_i64(x0));
                    atomicAdd((unsigned long long *)&gy[key], (unsigned long long)gg_fix_i64(x1));
                    atomicAdd((unsigned long long *)&gz[key], (unsigned long long)gg_fix_i64(x2));
                    atomicAdd(&gc[key], 1);
                } else {
                    // (key < 0: the flat row itself; key >= 0: the same row, flat_ = b*N - 1 + key)
                    atomicAdd(&p.fgs[flat_ * 4 + 0], a[u].y); atomicAdd(&p.fgs[flat_ * 4 + 1], a[u].z);
                    atomicAdd(&p.fgs[flat_ * 4 + 2], a[u].w); atomicAdd(&p.fgs[flat_ * 4 + 3], 1.f);
                }
                g9[0] += a[u].y * a[u].y; g9[1] += a[u].y * a[u].z; g9[2] += a[u].y * a[u].w;
                g9[3] += a[u].z * a[u].y; g9[4] += a[u].z * a[u].z; g9[5] += a[u].z * a[u].w;
                g9[6] += a[u].w * a[u].y; g9[7] += a[u].w * a[u].z; g9[8] += a[u].w * a[u].w;
                g9[9] += a[u].y; g9[10] += a[u].z; g9[11] += a[u].w;
            }
        }
        __syncthreads();
        float *gp_ = p.gpart + (((size_t)b * p.nsplit + sp) * (N + 1)) * 4;
        const float gi = ldexpf(1.f, -gk);
        for (int i = tid; i <= N; i += 1024)
            ((float4 *)gp_)[i] = make_float4((float)gx[i] * gi, (float)gy[i] * gi, (float)gz[i] * gi, (float)gc[i]);
    } else {
        for (int i = tid; i < (N + 1) * 16; i += 1024) acc[i] = 0;
    }
    // U centres per group and round.  amax -> nebidx[arg-max edge] -> LDS row is a dependent chain of
    // two memory levels; every level is issued for all U centres at once, and the first level of the
    // NEXT round is requested before the second level of this one is waited for
    constexpr int U = 8;
    struct L1 { unsigned pm[U]; float gv[U], zs[U]; };
    auto load1 = [&](int ob0, L1 &r) {
        size_t ob[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int o = ob0 + 64 * u < o1 ? ob0 + 64 * u : o0 + grp;
            ob[u] = (size_t)(b * O + o) * C + c;
            r.pm[u] = p.amax[ob[u]];
        }
#pragma unroll
        for (int u = 0; u < U; u++) { r.gv[u] = p.gval[ob[u]]; r.zs[u] = p.zsel[ob[u]]; }
    };
    const bool work = cok && o0 + grp < o1;
    L1 cur;
    float sm = 0.f;
    if (work) {
        load1(o0 + grp, cur);
        // scale of this channel's sums: the largest |scale * gradient| of the first round, whatever
        // the ReLU mask says (a channel whose first entries are all masked still gets a sane scale)
#pragma unroll
        for (int u = 0; u < U; u++) {
            const float v = fabsf(scv * cur.gv[u]);
            if (o0 + grp + 64 * u < o1 && v < 3.0e38f) sm = fmaxf(sm, v);
        }
    }
    __syncthreads();                                              // acc zeroed, cmax zeroed
    if (work) atomicMax(&cmax[lane], __float_as_uint(sm));
    __syncthreads();
    const int fk = gg_fix_exp(cmax[lane]);
    const float F = ldexpf(1.f, fk);
    const float thr = fminf(0x1p+47f, 0x1p+62f / ((float)per + 1.f));
    if (work) {
        for (int ob0 = o0 + grp; ob0 < o1; ob0 += 64 * U) {
            int idx[U];
            float4 av[U];
            // (edge numbers in 32 bits -- B*O*P < 2^31 is checked by the launcher: with 64-bit index
            // arithmetic on the loaded byte the compiler funnels all the amax loads through one
            // register pair and waits for each of them in turn)
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int o = ob0 + 64 * u < o1 ? ob0 + 64 * u : o0 + grp;
                const unsigned e = (unsigned)(b * O + o) * (unsigned)P + cur.pm[u];
                idx[u] = p.nebidx[e];
                av[u] = *(const float4 *)(p.att16 + (size_t)e * 16);
            }
            L1 nxt;
            load1(ob0 + 64 * U, nxt);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (ob0 + 64 * u >= o1) break;
                const float s = (cur.zs[u] * scv + shv > 0.f) ? scv * cur.gv[u] : 0.f;
                const int key = keyof(idx[u]);
                if (s != 0.f) {
                    const float x = s * F;
                    if (key >= 0 && fabsf(x) < thr)
                        atomicAdd((unsigned long long *)&acc[key * 16 + lane], (unsigned long long)gg_fix_i64(x));
                    else
                        atomicAdd(&p.fpart[flat_ * C + c], s);
                }
                wg0 = fmaf(av[u].y, s, wg0); wg1 = fmaf(av[u].z, s, wg1); wg2 = fmaf(av[u].w, s, wg2);
            }
            cur = nxt;
        }
    }
    __syncthreads();
    // flush: every element of the LDS copy is stored (the finish kernel sums the splits); element i
    // belongs to channel i & 15 = this thread's own (1024 is a multiple of 16), so to its scale
    float *pp_ = p.part + (((size_t)b * p.nsplit + sp) * (N + 1)) * C;
    const float Fi = ldexpf(1.f, -fk);
    if (cok)
        for (int i = tid; i < (N + 1) * 16; i += 1024)
            pp_[(size_t)(i >> 4) * C + c] = (float)acc[i] * Fi;
    // dWg sparse part: lanes hold channel c, 64 groups -> LDS -> one atomic per channel
    __syncthreads();
    float *wr = lds;                                   // [3][64][16]
    wr[(0 * 64 + grp) * 16 + lane] = wg0;
    wr[(1 * 64 + grp) * 16 + lane] = wg1;
    wr[(2 * 64 + grp) * 16 + lane] = wg2;
    __syncthreads();
    if (tid < 48) {
        const int j = tid >> 4, l = tid & 15;
        float v = 0.f;
        for (int g = 0; g < 64; g++) v += wr[(j * 64 + g) * 16 + l];
        if (c0 + l < C) atomicAdd(&p.wgs[j * C + c0 + l], (double)v);
    }
    if (geo_wg) {
#pragma unroll
        for (int i = 0; i < 12; i++) {
            float v = g9[i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if ((tid & 63) == 0) red[tid >> 6][i] = v;
        }
        __syncthreads();
        if (tid < 12) {
            float v = 0.f;
            for (int w = 0; w < 16; w++) v += red[w][tid];
            atomicAdd(&p.gg[tid], (double)v);
        }
    }
}

// dYsrc[r][c], Gsum[r][4] from the partials + the dense (affine) part
__global__ __launch_bounds__(256) void gg_k_edge_lin0_bwd_finish(
    const float *__restrict__ part, const float *__restrict__ gpart, const float *__restrict__ Ysrc,
    const float *__restrict__ Wg, const float *__restrict__ bias, const float *__restrict__ scale,
    const float *__restrict__ mean, const float *__restrict__ rstd, const float *__restrict__ m1,
    const float *__restrict__ m2, int B, int N, int C, int nsplit, float *__restrict__ dYsrc,
    float *__restrict__ Gsum, const float *__restrict__ fpart, const float *__restrict__ fgs,
    const float *__restrict__ Gin)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)B * N * C) return;
    const long long r = t / C;
    const int c = (int)(t - r * C);
    const int b = (int)(r / N), n = (int)(r - (long long)b * N);
    // destination row r collects key n+1 of its own cloud and, for its last row, key 0 of the next
    // (Gin: the per-source geo sums are the forward's)
    const float4 f4 = *(const float4 *)((Gin ? Gin : fgs) + r * 4);
    float s = fpart[t], g0 = f4.x, g1 = f4.y, g2 = f4.z, cnt = f4.w;
    for (int sp = 0; sp < nsplit; sp++) {
        const size_t base = ((size_t)b * nsplit + sp) * (N + 1) + (n + 1);
        s += part[base * C + c];
        if (!Gin) {
            const float4 g = *(const float4 *)(gpart + base * 4);
            g0 += g.x; g1 += g.y; g2 += g.z; cnt += g.w;
        }
        if (n == N - 1 && b + 1 < B) {
            const size_t nb_ = ((size_t)(b + 1) * nsplit + sp) * (N + 1);
            s += part[nb_ * C + c];
            if (!Gin) {
                const float4 h = *(const float4 *)(gpart + nb_ * 4);
                g0 += h.x; g1 += h.y; g2 += h.z; cnt += h.w;
            }
        }
    }
    const float sc = scale[c];
    const float bz = -(sc * rstd[c]) * m2[c], cz = -(sc * m1[c]);
    float lin = cnt * ((Ysrc ? Ysrc[r * C + c] : 0.f) + bias[c]);
    if (Wg) lin += g0 * Wg[c] + g1 * Wg[C + c] + g2 * Wg[2 * C + c];
    dYsrc[t] = s + (bz * lin + cnt * (cz - mean[c] * bz));
    if (c == 0 && !Gin) *(float4 *)(Gsum + r * 4) = make_float4(g0, g1, g2, cnt);
}

int gg_edge_lin0_sparse_nsplit(int B, int C);

size_t gg_edge_lin0_sparse_workspace(int B, int N, int C)
{
    const int nsplit = gg_edge_lin0_sparse_nsplit(B, C);
    return ((size_t)B * nsplit * (N + 1) * (C + 4) + (size_t)B * N * (C + 4)) * sizeof(float);
}

int gg_edge_lin0_sparse_nsplit(int B, int C)
{
    int ns = 512 / (B * ((C + 15) / 16));
    return ns < 1 ? 1 : (ns > 32 ? 32 : ns);
}

// 1 = shape not supported.  wgs[3*C] and gg[12] (fp64) must be zero-filled by the caller.
int gg_edge_lin0_bwd_sparse(const int *nebidx, const float *att16, const unsigned char *amax,
                            const float *gval, const float *zsel, const float *Ysrc,
                            const float *Wg, const float *bias, const float *scale,
                            const float *shift, const float *mean, const float *rstd,
                            const float *m1, const float *m2, int B, int N, int O, int P, int C,
                            float *dYsrc, float *Gsum, double *wgs, double *gg, void *workspace,
                            hipStream_t st, int geo_given)
{
    size_t lds = (size_t)(N + 1) * 16 * 8;          // acc[(N+1)][16] int64 (the geo sums alias its start)
    if (lds > 150 * 1024 || C < 1 || (C & 3) || (long long)B * O * P >= (1ll << 31)) return 1;
    if (lds < 3 * 64 * 16 * 4) lds = 3 * 64 * 16 * 4;
    static GGDevOnce attr_done;
    if (!attr_done) {
        // (the kernel also has 1 KB of static LDS: dynamic + static must stay within 160 KB)
        if (hipFuncSetAttribute((const void *)gg_k_edge_lin0_bwd_sparse, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess) return 3;
        attr_done = true;
    }
    GGEdgeSparse p;
    p.nebidx = nebidx; p.att16 = att16; p.amax = amax; p.gval = gval; p.zsel = zsel;
    p.sc = scale; p.sh = shift; p.B = B; p.N = N; p.O = O; p.P = P; p.C = C;
    p.nsplit = gg_edge_lin0_sparse_nsplit(B, C);
    p.part = (float *)workspace;
    p.gpart = p.part + (size_t)B * p.nsplit * (N + 1) * C;
    p.fpart = p.gpart + (size_t)B * p.nsplit * (N + 1) * 4;
    p.fgs = p.fpart + (size_t)B * N * C;
    p.wgs = wgs; p.gg = gg; p.geo_given = geo_given;
    if (hipMemsetAsync(p.fpart, 0, (size_t)B * N * (C + 4) * sizeof(float), st) != hipSuccess) return 3;
    gg_k_edge_lin0_bwd_sparse<<<dim3(p.nsplit, (C + 15) / 16 + (geo_given ? 0 : 1), B), 1024, lds, st>>>(p);
    const long long tot = (long long)B * N * C;
    gg_k_edge_lin0_bwd_finish<<<(int)((tot + 255) / 256), 256, 0, st>>>(
        p.part, p.gpart, Ysrc, Wg, bias, scale, mean, rstd, m1, m2, B, N, C, p.nsplit, dYsrc, Gsum,
        p.fpart, p.fgs, geo_given ? Gsum : nullptr);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

// ------------------------------------------------------------------------------------------
// Weight gradient of the geo_vec columns of the source-side first conv, from the pieces the sparse
// backward leaves behind (one launch instead of ~20 element-wise ops on [3][C] tensors):
//   dWg[j][c] = wgs[j][c]                                         arg-max entries
//             + bz_c (T[j][c] + Gtot_j b_c + sum_k GG[j][k] Wg[k][c])      dense part, z0 affine
//             + (cz_c - mean_c bz_c) Gtot_j
// T = Ysrc^T Gsum ([C][4], columns 0..2 used), gg = (GG[9], Gtot[3]), wgb = (Wg[3][C], b[C]).
// Written transposed into dW[c][j] with row stride ld (columns 0..2 of the layer's dW).
__global__ __launch_bounds__(256) void gg_k_edge_lin0_dwg(
    const double *__restrict__ wgs, const double *__restrict__ gg, const float *__restrict__ T,
    const float *__restrict__ wgb, const float *__restrict__ scale, const float *__restrict__ mean,
    const float *__restrict__ rstd, const float *__restrict__ m1, const float *__restrict__ m2, int C,
    float *__restrict__ dW, int ld)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float sc = scale[c];
    const float bz = -(sc * rstd[c]) * m2[c], cz = -(sc * m1[c]);
    const float w0 = wgb[c], w1 = wgb[C + c], w2 = wgb[2 * C + c], b = wgb[3 * C + c];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const float gt = (float)gg[9 + j];
        const float t1 = (T[c * 4 + j] + gt * b) +
                         (((float)gg[3 * j] * w0 + (float)gg[3 * j + 1] * w1) + (float)gg[3 * j + 2] * w2);
        dW[(size_t)c * ld + j] = (float)wgs[j * C + c] + bz * t1 + (cz - mean[c] * bz) * gt;
    }
}

int gg_edge_lin0_dwg(const double *wgs, const double *gg, const float *T, const float *wgb,
                     const float *scale, const float *mean, const float *rstd, const float *m1,
                     const float *m2, int C, float *dW, int ld, hipStream_t st)
{
    gg_k_edge_lin0_dwg<<<(C + 255) / 256, 256, 0, st>>>(wgs, gg, T, wgb, scale, mean, rstd, m1, m2, C,
                                                       dW, ld);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
