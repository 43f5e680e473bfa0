// driver.cpp -- extern "C" entries of the host-side emulation of the index operators (tests/simt/simt_hip.h).
// Linked with the kernel sources of grid_gcn_amd/csrc as rewritten by tests/simt/build.py (launch syntax only), one
// translation unit each as in the product's build; here: the same orchestration as grid_gcn_amd/csrc/gridgcn_capi.hip
// (gridify_common / gridgcn_gridify_up / gridgcn_ball_knn*), statement for statement, so that what runs here is the
// product's code path minus the GPU.  TEST INFRASTRUCTURE: nothing under grid_gcn_amd/ knows this file.
#include "simt_hip.h"

#include "gridgcn_fillgrid.h"
#include "gridgcn_index.h"
#include "gridgcn_train.h"

// the launchers of the kernel translation units (declared as gridgcn_capi.hip declares them)
int gg_launch_query_gridify(const float *data, int B, int N, const GGGrid &gp, char *wsbase, const GGIndexWs &w,
                            int *nebidx, float *nebmsk, float *cent, float *centmsk, const int *centnum, hipStream_t st);
int gg_launch_query_knn(const float *data, int B, int N, const GGGrid &gp, char *wsbase, const GGIndexWs &w, int *nebidx,
                        float *nebmsk, float *cent, float *centmsk, const int *centnum, hipStream_t st);
int gg_launch_query_up(const float *updata, const int *up_np, int B, int Nd, const GGGrid &gp, char *wsbase,
                       const GGIndexWs &w, int *nebidx, float *nebmsk, hipStream_t st);
int gg_ball_knn(const float *, const float *, const int *, const int *, int, int, int, int, float, int *, hipStream_t,
                int su = 3, int sk = 3, int ztail = 0);
int gg_knn(const float *, const float *, const int *, const int *, int, int, int, int, int *, hipStream_t);
size_t gg_ball_grid_workspace(int B, int m);
int gg_ball_knn_grid(const float *, const float *, const int *, const int *, int, int, int, int, float, int *, void *,
                     hipStream_t, int su = 3, int sk = 3, int ztail = 0);
bool gg_att_bwd_noz_ok(long long E, int cin, int C);
bool gg_att_bwd_noz_mom_ok(long long E, int cin, int C, int P);
size_t gg_att_bwd_noz_workspace(long long E);
int gg_att_bwd_noz(const float *, const float *, const float *, const float *, const float *, const float *,
                   const float *, const float *, const float *, const float *, const double *, const unsigned char *,
                   const float *, int, long long, float *, float *, float *, float *, float *, float *, double *,
                   double *, void *, hipStream_t, const double *mom = nullptr);
void gg_set_att_nz_v2(int v);
int gg_att_moments_grid_of(long long E);
extern long long simt_buf_oob;

static size_t ws_align(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" {

void simt_counters(long long *out)
{
    out[0] = simt_launches;
    out[1] = simt_rendezvous;
    out[2] = simt_foreign_reads;
    out[3] = simt_divergent_rendezvous;
    out[4] = simt_buf_oob;
}

void simt_set_option(int which, int value) { gg_index_set_tuning(which, value); }

// mode 0: Gridify, 1: GridifyKNN, 2: Gridify_occaware (beta), 3: fast_rand
int simt_gridify_workspace_bytes(int mode, int B, int N, const gridgcn_grid_params *p, size_t *bytes)
{
    GGGrid gp;
    int rc = gg_fill_grid(p, B, N, mode == 3, &gp);
    if (rc || !bytes) return rc ? rc : GRIDGCN_EINVAL;
    size_t need = gg_index_workspace_bytes(B, N, gp, true, nullptr);
    if (mode >= 2) need = ws_align(need) + gg_cas_workspace_bytes(B, N, gp);
    *bytes = need;
    return GRIDGCN_OK;
}

int simt_gridify(int mode, const float *data, const int32_t *np, int B, int N, const gridgcn_grid_params *p, float beta,
                 int32_t *nebidx, float *nebmsk, float *cent, float *centmsk, int32_t *centnum, void *ws,
                 size_t ws_bytes)
{
    GGGrid gp;
    int rc = gg_fill_grid(p, B, N, mode == 3, &gp);
    if (rc) return rc;
    GGIndexWs w;
    size_t need = gg_index_workspace_bytes(B, N, gp, true, &w);
    const size_t cas_off = ws_align(need);
    if (mode >= 2) need = cas_off + gg_cas_workspace_bytes(B, N, gp);
    if (!ws || ws_bytes < need) return GRIDGCN_EWORKSPACE;
    if (gg_index_init()) return GRIDGCN_ELAUNCH;
    hipStream_t st = nullptr;
    rc = gg_index_build(data, np, B, N, gp, true, centnum, (char *)ws, w, st);
    if (rc) return rc;
    if (mode == 3)      // gridgcn_gridify_fast_rand: the variant's own slots + query
        return gg_fastrand_query(data, np, B, N, gp, (char *)ws, w, (char *)ws + cas_off, nebidx, nebmsk, cent,
                                 centmsk, centnum, st);
    if (mode == 2) {
        rc = gg_cas_refine(data, np, B, N, gp, beta, (int *)((char *)ws + w.o_slotfirst1), centnum,
                           (const int2 *)((char *)ws + w.o_vtab), (const int *)((char *)ws + w.o_sorted),
                           (char *)ws + cas_off, st);
        if (rc) return rc;
    }
    if (mode == 1)
        return gg_launch_query_knn(data, B, N, gp, (char *)ws, w, nebidx, nebmsk, cent, centmsk, centnum, st);
    return gg_launch_query_gridify(data, B, N, gp, (char *)ws, w, nebidx, nebmsk, cent, centmsk, centnum, st);
}

int simt_gridify_up_workspace_bytes(int B, int Nd, const gridgcn_grid_params *p, size_t *bytes)
{
    GGGrid gp;
    int rc = gg_fill_grid(p, B, Nd, true, &gp);
    if (rc || !bytes) return rc ? rc : GRIDGCN_EINVAL;
    *bytes = gg_index_workspace_bytes(B, Nd, gp, false, nullptr);
    return GRIDGCN_OK;
}

int simt_gridify_up(const float *downdata, const float *updata, const int32_t *down_np, const int32_t *up_np, int B,
                    int Nd, const gridgcn_grid_params *p, int32_t *nebidx, float *nebmsk, void *ws, size_t ws_bytes)
{
    GGGrid gp;
    int rc = gg_fill_grid(p, B, Nd, true, &gp);
    if (rc) return rc;
    GGIndexWs w;
    size_t need = gg_index_workspace_bytes(B, Nd, gp, false, &w);
    if (!ws || ws_bytes < need) return GRIDGCN_EWORKSPACE;
    if (gg_index_init()) return GRIDGCN_ELAUNCH;
    rc = gg_index_build(downdata, down_np, B, Nd, gp, false, nullptr, (char *)ws, w, nullptr);
    if (rc) return rc;
    return gg_launch_query_up(updata, up_np, B, Nd, gp, (char *)ws, w, nebidx, nebmsk, nullptr);
}

int simt_ball_knn(const float *unknown, const float *known, const int32_t *downnum, const int32_t *upnum, int B,
                  int n, int m, int k, float radius, int32_t *idx)
{
    if (B < 1 || n < 1 || m < 1 || k < 1 || k > 6) return GRIDGCN_EINVAL;
    return gg_ball_knn(unknown, known, downnum, upnum, B, n, m, k, radius, idx, nullptr, 3, 3, 0);
}

int simt_knn_all(const float *unknown, const float *known, const int32_t *downnum, const int32_t *upnum, int B, int n,
                 int m, int k, int32_t *idx)
{
    if (B < 1 || n < 1 || m < 1 || k < 1 || k > 6) return GRIDGCN_EINVAL;
    return gg_knn(unknown, known, downnum, upnum, B, n, m, k, idx, nullptr);
}

size_t simt_ball_grid_workspace(int B, int m) { return gg_ball_grid_workspace(B, m); }

int simt_ball_knn_grid(const float *unknown, const float *known, const int32_t *downnum, const int32_t *upnum,
                       int B, int n, int m, int k, float radius, int32_t *idx, void *ws)
{
    if (B < 1 || n < 1 || m < 1 || k < 1 || k > 6) return GRIDGCN_EINVAL;
    return gg_ball_knn_grid(unknown, known, downnum, upnum, B, n, m, k, radius, idx, ws, nullptr, 3, 3,
                                           0);
}

// ---- training kernels (the entries of gridgcn_capi.hip, argument checks included) --------------------------------
size_t simt_att_bwd_noz_workspace(long long E) { return gg_att_bwd_noz_workspace(E); }
size_t simt_att_moments_workspace(long long E) { return gg_att_moments_workspace(E); }
size_t simt_att_moments_offset(long long E) { return (size_t)gg_att_moments_grid_of(E) * 17 * 64 * sizeof(double); }
void simt_set_att_nz_v2(int v) { gg_set_att_nz_v2(v); }

int simt_att_bn2_moments(const float *Z1, const float *scale1, const float *shift1, const float *W2, const float *b2,
                         const float *gamma, const float *beta, long long E, float eps, float momentum, float *scale,
                         float *shift, float *mean, float *rstd, double *sums, void *ws)
{
    return gg_att_bn2_moments(Z1, scale1, shift1, W2, b2, gamma, beta, E, eps, momentum, scale, shift, mean, rstd,
                              nullptr, nullptr, nullptr, sums, ws, nullptr);
}

// moments == NULL: gridgcn_att_bwd_noz; else gridgcn_att_bwd_noz_mom
int simt_att_bwd_noz(const float *Z1, const float *pscale, const float *pshift, const float *pmean, const float *prstd,
                     const float *W2, const float *b2, const float *scale, const float *mean, const float *rstd,
                     const double *sums, const uint8_t *amax, const float *gval, int P, long long E,
                     const double *moments, float *dX, float *dW, float *m1, float *m2, float *dgamma, float *dbeta,
                     double *psums, double *s1, void *ws)
{
    if (!gg_att_bwd_noz_ok(E, 32, 128) || P < 1 || P > 256 || (E % P)) return GRIDGCN_EINVAL;
    if (moments && !gg_att_bwd_noz_mom_ok(E, 32, 128, P)) return GRIDGCN_EINVAL;
    const int rc = gg_att_bwd_noz(Z1, pscale, pshift, pmean, prstd, W2, b2, scale, mean, rstd, sums, amax, gval, P, E,
                                  dX, dW, m1, m2, dgamma, dbeta, psums, s1, ws, nullptr, moments);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int simt_att_pairmax_fwd(const float *Ysrc, const int32_t *nebidx, const float *att16, const float *Wg, const float *b,
                         int B, int Nsrc, int O, const float *Z1, const float *scale1, const float *shift1,
                         const float *W2, const float *b2, const float *scale_p, const float *shift_p,
                         const float *scale_a, const float *shift_a, long long ncent, float *agg, int ld_agg,
                         uint8_t *amax, float *zsel)
{
    if (!gg_att_fwd_ok(ncent, O, 5, 32, 128, ld_agg, (long long)B * Nsrc)) return GRIDGCN_EINVAL;
    return gg_att_pairmax_args(Ysrc, nebidx, att16, Wg, b, B, Nsrc, O, Z1, scale1, shift1, W2, b2, scale_p, shift_p,
                               scale_a, shift_a, ncent, agg, ld_agg, amax, zsel, nullptr);
}

}  // extern "C"
