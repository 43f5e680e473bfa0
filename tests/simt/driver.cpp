// driver.cpp -- extern "C" entries of the host-side emulation of the index operators (tests/simt/simt_hip.h).
// One translation unit: the emulator header, then the kernel sources of grid_gcn_amd/csrc as rewritten by
// tests/simt/build.py (launch syntax only), then the same orchestration as grid_gcn_amd/csrc/gridgcn_capi.hip
// (gridify_common / gridgcn_gridify_up / gridgcn_ball_knn*), statement for statement, so that what runs here is the
// product's code path minus the GPU.  TEST INFRASTRUCTURE: nothing under grid_gcn_amd/ knows this file.
#include "simt_hip.h"

#include "gridgcn_fillgrid.h"
#include "gridgcn_index.h"

#include "gridgcn_index.simt.inc"
#include "gridgcn_index_legacy.simt.inc"
#include "gridgcn_query.simt.inc"
#include "gridgcn_query_knn.simt.inc"
#include "gridgcn_fastrand.simt.inc"
#include "gridgcn_cas.simt.inc"
namespace simt_knn {
#include "gridgcn_knn.simt.inc"
}
namespace simt_ballgrid {
#include "gridgcn_ballgrid.simt.inc"
}

static size_t ws_align(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" {

void simt_counters(long long *out)
{
    out[0] = simt_launches;
    out[1] = simt_rendezvous;
    out[2] = simt_foreign_reads;
    out[3] = simt_divergent_rendezvous;
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
    return simt_knn::gg_ball_knn(unknown, known, downnum, upnum, B, n, m, k, radius, idx, nullptr, 3, 3, 0);
}

int simt_knn_all(const float *unknown, const float *known, const int32_t *downnum, const int32_t *upnum, int B, int n,
                 int m, int k, int32_t *idx)
{
    if (B < 1 || n < 1 || m < 1 || k < 1 || k > 6) return GRIDGCN_EINVAL;
    return simt_knn::gg_knn(unknown, known, downnum, upnum, B, n, m, k, idx, nullptr);
}

size_t simt_ball_grid_workspace(int B, int m) { return simt_ballgrid::gg_ball_grid_workspace(B, m); }

int simt_ball_knn_grid(const float *unknown, const float *known, const int32_t *downnum, const int32_t *upnum,
                       int B, int n, int m, int k, float radius, int32_t *idx, void *ws)
{
    if (B < 1 || n < 1 || m < 1 || k < 1 || k > 6) return GRIDGCN_EINVAL;
    return simt_ballgrid::gg_ball_knn_grid(unknown, known, downnum, upnum, B, n, m, k, radius, idx, ws, nullptr, 3, 3,
                                           0);
}

}  // extern "C"
