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
#include "gridgcn_edgelin.h"

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
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        const bool rw = !p.Z && live;
        w0[i] = (rw && p.Wg) ? p.Wg[cl + i] : 0.f;
        w1[i] = (rw && p.Wg) ? p.Wg[C0 + cl + i] : 0.f;
        w2[i] = (rw && p.Wg) ? p.Wg[2 * C0 + cl + i] : 0.f;
        bb[i] = rw ? p.b[cl + i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        const float s = live ? p.scale[cl + i] : 0.f;
        sc[i] = s; sh[i] = live ? p.shift[cl + i] : 0.f; mu[i] = live ? p.mean[cl + i] : 0.f;
        bz[i] = live ? -(s * p.rstd[cl + i]) * p.m2[cl + i] : 0.f;
        cz[i] = live ? -(s * p.m1[cl + i]) : 0.f;
        wg[0][i] = 0.f; wg[1][i] = 0.f; wg[2][i] = 0.f; acc[i] = 0.f;
    }
    const bool active = wid < p.B * p.cpc;
    if (active) {
        const int b = wid / p.cpc, ch = wid - b * p.cpc;
        const int e0 = ch * GG_CSR_CHUNK, e1 = e0 + GG_CSR_CHUNK < M ? e0 + GG_CSR_CHUNK : M;
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
            const bool whole = key != 0 && key != N && rbeg == rp[key] && rend == rp[key + 1];
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
    p.cpc = (p.M + GG_CSR_CHUNK - 1) / GG_CSR_CHUNK;
    const int nwave = p.B * p.cpc, grid = (nwave + 3) / 4;
    if (VPL == 1) gg_k_edge_lin0_bwd<1><<<grid, 256, 0, st>>>(p);
    else if (VPL == 2) gg_k_edge_lin0_bwd<2><<<grid, 256, 0, st>>>(p);
    else gg_k_edge_lin0_bwd<4><<<grid, 256, 0, st>>>(p);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
