// gridgcn_capi.hip -- the extern "C" boundary declared in include/gridgcn.h.
// Argument checking mirrors the reference's CHECK_EQs (gridify-inl.h:174-182,
// ball_k_nn.cc:34-42) and adds the limits this implementation relies on.
#include "../../include/gridgcn.h"
#include "gridgcn_index.h"
#include "gridgcn_fillgrid.h"
#include "gridgcn_conv.h"
#include "gridgcn_train.h"
#include "gridgcn_csr.h"
#include "gridgcn_edgelin.h"
#include "gridgcn_atteval.h"
#include "gridgcn_clsblock.h"
#include "gridgcn_optim.h"

int gg_launch_query_gridify(const float *data, int B, int N, const GGGrid &gp, char *wsbase,
                            const GGIndexWs &w, int *nebidx, float *nebmsk, float *cent,
                            float *centmsk, const int *centnum, hipStream_t st);
int gg_launch_query_knn(const float *data, int B, int N, const GGGrid &gp, char *wsbase,
                        const GGIndexWs &w, int *nebidx, float *nebmsk, float *cent,
                        float *centmsk, const int *centnum, hipStream_t st);
int gg_launch_query_up(const float *updata, const int *up_np, int B, int Nd, const GGGrid &gp,
                       char *wsbase, const GGIndexWs &w, int *nebidx, float *nebmsk,
                       hipStream_t st);
int gg_ball_knn(const float *, const float *, const int *, const int *, int, int, int, int, float,
                int *, hipStream_t, int su = 3, int sk = 3, int ztail = 0);
int gg_knn(const float *, const float *, const int *, const int *, int, int, int, int, int *,
           hipStream_t);
int gg_batch_take(const float *, const int *, int, int, int, int, float *, hipStream_t);
int gg_batch_take_backward(const float *, const int *, int, int, int, int, float *, int, int,
                           hipStream_t);
int gg_edge_inputs(const float *, const int *, const float *, int, int, int, int, int, int, int, int,
                   float *, float *, hipStream_t);

int gg_edge_inputs_rows(const float *, const int *, const float *, int, int, int, int, int, int, int,
                        int, int, float *, float *, hipStream_t);
int gg_ce_fwd(const float *, int, int, const long long *, long long, int, float *, double *, float *,
              hipStream_t);
int gg_ce_bwd(const float *, int, int, const long long *, long long, int, const float *,
              const double *, const float *, const float *, float *, hipStream_t);
int gg_colsum(const float *, long long, int, int, double *, float *, hipStream_t);
size_t gg_ball_grid_workspace(int B, int m);
int gg_ball_knn_grid(const float *, const float *, const int *, const int *, int, int, int, int,
                     float, int *, void *, hipStream_t, int su = 3, int sk = 3, int ztail = 0);
extern int gg_pairmax_split;        // gridgcn_pairmax.hip (GRIDGCN_OPT_PAIRMAX_SPLIT)
int gg_cat_mask(const float *a, int lda, int ca, const float *b, int ldb, int cb, const float *mask,
                float *out, int ldo, float *out2, int ldo2, long long E, hipStream_t st);
int gg_mask_sum(const float *g1, int ld1, const float *g2, int ld2, int col0, int C, const float *mask,
                float *out, long long E, hipStream_t st);
size_t gg_take_bwd_sorted_workspace(int B, int N, int M);
int gg_take_bwd_sorted(const float *, const int *, int, int, int, int, float *, int, int, void *,
                       hipStream_t);
int gg_pairmax_fwd(const float *, const float *, const float *, const float *, const float *,
                   const float *, long long, int, int, float *, int, unsigned char *, float *,
                   hipStream_t);
int gg_pairmax_fwd_src(const float *, const int *, const float *, const float *, const float *, int,
                       int, int, const float *, const float *, const float *, const float *,
                       const float *, long long, int, int, float *, int, unsigned char *, float *,
                       int, hipStream_t);
int gg_pairmax_bwd(const float *, const float *, const float *, const float *, const float *,
                   const float *, const float *, const float *, const float *, const float *,
                   const float *, const unsigned char *, long long, int, int, int, float *, float *,
                   double *, double *, const float *, int, hipStream_t);

int gg_pack_linear(const float *, const float *, int, int, int, int, int, float *, float *,
                   float *, float *, float *, float *, hipStream_t);
int gg_bn_finalize(const double *, const float *, const float *, long long, float, float, int,
                   float *, float *, float *, float *, float *, float *, long long *, hipStream_t, int tail = 0);
int gg_gemm_small(int mode, const float *A, int lda, const float *B, int ldb, float *C, int ldc, int M, int N,
                  int K, int zero_left, const float *bias, void *ws, hipStream_t st);      // gridgcn_gemm.hip
size_t gg_gemm_small_workspace(int M, int N, int K);
int gg_bn_bwd_finalize(const double *, long long, int, float *, float *, float *, float *,
                       hipStream_t);

bool gg_att_bwd_noz_ok(long long E, int cin, int C);               // gridgcn_attbwd_nz.hip
void gg_set_att_nz_v2(int v);
int gg_get_att_nz_v2();
size_t gg_att_bwd_noz_workspace(long long E);
int gg_att_bwd_noz(const float *, const float *, const float *, const float *, const float *, const float *,
                   const float *, const float *, const float *, const float *, const double *,
                   const unsigned char *, const float *, int, long long, float *, float *, float *, float *,
                   float *, float *, double *, double *, void *, hipStream_t, const double *mom = nullptr);
bool gg_att_bwd_noz_mom_ok(long long E, int cin, int C, int P);
int gg_att_moments_grid_of(long long E);                           // gridgcn_attfwd.hip
int gg_pack_desc_fill(gridgcn_pack_desc *e);
int gg_pack_linear_batch(const gridgcn_pack_desc *dev, int nlayers, int max_n, hipStream_t st);

static int ensure_init()
{
    static int rc = gg_index_init();  // thread-safe one-time init (C++11 static)
    return rc;
}

static int fill_grid(const gridgcn_grid_params *p, int B, int N, bool up, GGGrid *gp)
{
    return gg_fill_grid(p, B, N, up, gp);
}

extern "C" {

const char *gridgcn_strerror(int code)
{
    switch (code) {
    case GRIDGCN_OK: return "ok";
    case GRIDGCN_EINVAL: return "invalid argument (shape/attribute outside the supported domain)";
    case GRIDGCN_EWORKSPACE: return "workspace missing or too small";
    case GRIDGCN_ELAUNCH: return "HIP launch failed";
    default: return "unknown error";
    }
}

int gridgcn_abi_version(void) { return 9; }

int gridgcn_set_option(int option, int value)
{
    if (option == GRIDGCN_OPT_ATT_BWD_FUSED) { gg_set_att_bwd_fused(value); return GRIDGCN_OK; }
    if (option == GRIDGCN_OPT_INDEX_SLAB_SHIFT) {
        if (value < -4 || value > 4) return GRIDGCN_EINVAL;
        gg_index_set_tuning(0, value);
        return GRIDGCN_OK;
    }
    if (option == GRIDGCN_OPT_INDEX_CHUNK) {
        if (value != 0 && value != 1024 && value != 2048 && value != 4096) return GRIDGCN_EINVAL;
        gg_index_set_tuning(1, value);
        return GRIDGCN_OK;
    }
    if (option == GRIDGCN_OPT_INDEX_SMALL) {
        if (value != 0 && value != 1) return GRIDGCN_EINVAL;
        gg_index_set_tuning(2, value);
        return GRIDGCN_OK;
    }
    if (option == GRIDGCN_OPT_COL_SPLIT) {
        if (value != 0 && value != 1) return GRIDGCN_EINVAL;
        gg_set_col_split(value);
        return GRIDGCN_OK;
    }
    if (option == GRIDGCN_OPT_PAIRMAX_SPLIT) {
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return GRIDGCN_EINVAL;
        gg_pairmax_split = value;
        return GRIDGCN_OK;
    }
    if (option == GRIDGCN_OPT_ATT_NZ_V2) {
        if (value != 0 && value != 1) return GRIDGCN_EINVAL;
        gg_set_att_nz_v2(value);
        return GRIDGCN_OK;
    }
    if (option == GRIDGCN_OPT_BWD_FUSED128) {
        if (value < 0 || value > 2) return GRIDGCN_EINVAL;
        gg_set_bwd_fused128(value);
        return GRIDGCN_OK;
    }
    if (option == GRIDGCN_OPT_ATT_EVAL_TILE) {
        if (value != 0 && value != 1) return GRIDGCN_EINVAL;
        gg_set_att_eval_tile(value);
        return GRIDGCN_OK;
    }
    return GRIDGCN_EINVAL;
}

int gridgcn_get_option(int option)
{
    if (option == GRIDGCN_OPT_ATT_BWD_FUSED) return gg_get_att_bwd_fused();
    if (option == GRIDGCN_OPT_INDEX_SLAB_SHIFT) return gg_index_get_tuning(0);
    if (option == GRIDGCN_OPT_INDEX_CHUNK) return gg_index_get_tuning(1);
    if (option == GRIDGCN_OPT_INDEX_SMALL) return gg_index_get_tuning(2);
    if (option == GRIDGCN_OPT_COL_SPLIT) return gg_get_col_split();
    if (option == GRIDGCN_OPT_PAIRMAX_SPLIT) return gg_pairmax_split;
    if (option == GRIDGCN_OPT_ATT_NZ_V2) return gg_get_att_nz_v2();
    if (option == GRIDGCN_OPT_BWD_FUSED128) return gg_get_bwd_fused128();
    if (option == GRIDGCN_OPT_ATT_EVAL_TILE) return gg_get_att_eval_tile();
    return -1;
}

int gridgcn_set_mlp_precision(int bf16)
{
    gg_set_mlp_bf16(bf16);
    return GRIDGCN_OK;
}

int gridgcn_get_mlp_precision(void) { return gg_get_mlp_bf16(); }

int gridgcn_gridify_workspace_bytes(int B, int N, const gridgcn_grid_params *p, size_t *bytes)
{
    GGGrid gp;
    int rc = fill_grid(p, B, N, false, &gp);
    if (rc || !bytes) return rc ? rc : GRIDGCN_EINVAL;
    *bytes = gg_index_workspace_bytes(B, N, gp, true, nullptr);
    return GRIDGCN_OK;
}

static size_t ws_align(size_t x) { return (x + 255) & ~(size_t)255; }

// cas_beta < 0: random voxel sampling (the reference's source); >= 0: coverage-aware refinement
static int gridify_common(bool knn, const float *data, const int32_t *np, int B, int N,
                          const gridgcn_grid_params *p, int32_t *nebidx, float *nebmsk,
                          float *cent, float *centmsk, int32_t *centnum, void *ws,
                          size_t ws_bytes, void *stream, float cas_beta = -1.0f)
{
    GGGrid gp;
    int rc = fill_grid(p, B, N, false, &gp);
    if (rc) return rc;
    if (!data || !np || !nebidx || !nebmsk || !cent || !centmsk || !centnum) return GRIDGCN_EINVAL;
    GGIndexWs w;
    size_t need = gg_index_workspace_bytes(B, N, gp, true, &w);
    const size_t cas_off = ws_align(need);
    if (cas_beta >= 0.0f) need = cas_off + gg_cas_workspace_bytes(B, N, gp);
    if (!ws || ws_bytes < need) return GRIDGCN_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (ensure_init()) return GRIDGCN_ELAUNCH;
    rc = gg_index_build(data, np, B, N, gp, true, centnum, (char *)ws, w, st);
    if (rc) return rc;
    if (cas_beta >= 0.0f) {
        rc = gg_cas_refine(data, np, B, N, gp, cas_beta, (int *)((char *)ws + w.o_slotfirst1),
                           centnum, (const int2 *)((char *)ws + w.o_vtab), (const int *)((char *)ws + w.o_sorted),
                           (char *)ws + cas_off, st);
        if (rc) return rc;
    }
    if (knn)
        return gg_launch_query_knn(data, B, N, gp, (char *)ws, w, nebidx, nebmsk, cent, centmsk,
                                   centnum, st);
    return gg_launch_query_gridify(data, B, N, gp, (char *)ws, w, nebidx, nebmsk, cent, centmsk,
                                   centnum, st);
}

int gridgcn_gridify(const float *data, const int32_t *np, int B, int N,
                    const gridgcn_grid_params *p, int32_t *nebidx, float *nebmsk, float *cent,
                    float *centmsk, int32_t *centnum, void *ws, size_t ws_bytes, void *stream)
{
    return gridify_common(false, data, np, B, N, p, nebidx, nebmsk, cent, centmsk, centnum, ws,
                          ws_bytes, stream);
}

int gridgcn_gridify_occaware_workspace_bytes(int B, int N, const gridgcn_grid_params *p,
                                             size_t *bytes)
{
    GGGrid gp;
    int rc = fill_grid(p, B, N, false, &gp);
    if (rc || !bytes) return rc ? rc : GRIDGCN_EINVAL;
    *bytes = ws_align(gg_index_workspace_bytes(B, N, gp, true, nullptr)) +
             gg_cas_workspace_bytes(B, N, gp);
    return GRIDGCN_OK;
}

int gridgcn_gridify_occaware(const float *data, const int32_t *np, int B, int N,
                             const gridgcn_grid_params *p, float beta, int32_t *nebidx,
                             float *nebmsk, float *cent, float *centmsk, int32_t *centnum, void *ws,
                             size_t ws_bytes, void *stream)
{
    if (!(beta >= 0.0f)) return GRIDGCN_EINVAL;
    if (p && p->max_o_grid > 16384) return GRIDGCN_EINVAL;
    if (p && (p->grid_size[0] > 1023 || p->grid_size[1] > 1023 || p->grid_size[2] > 1023))
        return GRIDGCN_EINVAL;   // the sweep packs voxel coordinates into 10 bits each
    return gridify_common(false, data, np, B, N, p, nebidx, nebmsk, cent, centmsk, centnum, ws,
                          ws_bytes, stream, beta);
}

int gridgcn_gridify_fast_rand_workspace_bytes(int B, int N, const gridgcn_grid_params *p,
                                              size_t *bytes)
{
    return gridgcn_gridify_occaware_workspace_bytes(B, N, p, bytes);
}

int gridgcn_gridify_fast_rand(const float *data, const int32_t *np, int B, int N,
                              const gridgcn_grid_params *p, int32_t *nebidx, float *nebmsk,
                              float *cent, float *centmsk, int32_t *centnum, void *ws,
                              size_t ws_bytes, void *stream)
{
    GGGrid gp;
    int rc = fill_grid(p, B, N, true, &gp);      // (B*N*k^3 < 2^31: the variant's int thread index)
    if (rc) return rc;
    if (!data || !np || !nebidx || !nebmsk || !cent || !centmsk || !centnum) return GRIDGCN_EINVAL;
    GGIndexWs w;
    const size_t need0 = gg_index_workspace_bytes(B, N, gp, true, &w);
    const size_t off = ws_align(need0);
    if (!ws || ws_bytes < off + gg_cas_workspace_bytes(B, N, gp)) return GRIDGCN_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (ensure_init()) return GRIDGCN_ELAUNCH;
    rc = gg_index_build(data, np, B, N, gp, true, centnum, (char *)ws, w, st);
    if (rc) return rc;
    return gg_fastrand_query(data, np, B, N, gp, (char *)ws, w, (char *)ws + off, nebidx, nebmsk,
                             cent, centmsk, centnum, st);
}

int gridgcn_gridify_timed(const float *data, const int32_t *np, int B, int N,
                          const gridgcn_grid_params *p, int32_t *nebidx, float *nebmsk, float *cent,
                          float *centmsk, int32_t *centnum, void *ws, size_t ws_bytes, void *stream,
                          int iters, float *ms_per_call)
{
    if (iters < 1 || !ms_per_call) return GRIDGCN_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess) return GRIDGCN_ELAUNCH;
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return GRIDGCN_ELAUNCH; }
    int rc = GRIDGCN_OK;
    for (int w = 0; w < 3 && !rc; w++)
        rc = gridify_common(false, data, np, B, N, p, nebidx, nebmsk, cent, centmsk, centnum, ws,
                            ws_bytes, stream);
    if (!rc && hipEventRecord(e0, st) != hipSuccess) rc = GRIDGCN_ELAUNCH;
    for (int i = 0; i < iters && !rc; i++)
        rc = gridify_common(false, data, np, B, N, p, nebidx, nebmsk, cent, centmsk, centnum, ws,
                            ws_bytes, stream);
    if (!rc && hipEventRecord(e1, st) != hipSuccess) rc = GRIDGCN_ELAUNCH;
    // (the stream is drained even after an error so that the events can be destroyed)
    if (hipStreamSynchronize(st) != hipSuccess && !rc) rc = GRIDGCN_ELAUNCH;
    float ms = 0.0f;
    if (!rc && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = GRIDGCN_ELAUNCH;
    *ms_per_call = rc ? 0.0f : ms / (float)iters;
    if (hipEventDestroy(e0) != hipSuccess && !rc) rc = GRIDGCN_ELAUNCH;
    if (hipEventDestroy(e1) != hipSuccess && !rc) rc = GRIDGCN_ELAUNCH;
    return rc;
}

int gridgcn_gridify_knn_workspace_bytes(int B, int N, const gridgcn_grid_params *p, size_t *bytes)
{
    return gridgcn_gridify_workspace_bytes(B, N, p, bytes);
}

int gridgcn_gridify_knn(const float *data, const int32_t *np, int B, int N,
                        const gridgcn_grid_params *p, int32_t *nebidx, float *nebmsk, float *cent,
                        float *centmsk, int32_t *centnum, void *ws, size_t ws_bytes, void *stream)
{
    return gridify_common(true, data, np, B, N, p, nebidx, nebmsk, cent, centmsk, centnum, ws,
                          ws_bytes, stream);
}

int gridgcn_gridify_up_workspace_bytes(int B, int Nd, const gridgcn_grid_params *p, size_t *bytes)
{
    GGGrid gp;
    int rc = fill_grid(p, B, Nd, true, &gp);
    if (rc || !bytes) return rc ? rc : GRIDGCN_EINVAL;
    *bytes = gg_index_workspace_bytes(B, Nd, gp, false, nullptr);
    return GRIDGCN_OK;
}

int gridgcn_gridify_up(const float *downdata, const float *updata, const int32_t *down_np,
                       const int32_t *up_np, int B, int Nd, const gridgcn_grid_params *p,
                       int32_t *nebidx, float *nebmsk, void *ws, size_t ws_bytes, void *stream)
{
    GGGrid gp;
    int rc = fill_grid(p, B, Nd, true, &gp);
    if (rc) return rc;
    if (!downdata || !updata || !down_np || !up_np || !nebidx || !nebmsk) return GRIDGCN_EINVAL;
    GGIndexWs w;
    size_t need = gg_index_workspace_bytes(B, Nd, gp, false, &w);
    if (!ws || ws_bytes < need) return GRIDGCN_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (ensure_init()) return GRIDGCN_ELAUNCH;
    rc = gg_index_build(downdata, down_np, B, Nd, gp, false, nullptr, (char *)ws, w, st);
    if (rc) return rc;
    return gg_launch_query_up(updata, up_np, B, Nd, gp, (char *)ws, w, nebidx, nebmsk, st);
}

int gridgcn_ball_knn(const float *unknown, const float *known, const int32_t *downnum,
                     const int32_t *upnum, int B, int n, int m, int k, float radius, int32_t *idx,
                     void *stream)
{
    if (!unknown || !known || !downnum || !upnum || !idx) return GRIDGCN_EINVAL;
    if (B < 1 || n < 1 || m < 1 || k < 1 || k > 6) return GRIDGCN_EINVAL;  // best[6]
    return gg_ball_knn(unknown, known, downnum, upnum, B, n, m, k, radius, idx,
                       (hipStream_t)stream);
}

int gridgcn_ball_knn_ld(const float *unknown, int ldu, const float *known, int ldk,
                        const int32_t *downnum, const int32_t *upnum, int B, int n, int m, int k,
                        float radius, int zero_tail, int32_t *idx, void *stream)
{
    if (!unknown || !known || !downnum || !upnum || !idx || ldu < 3 || ldk < 3) return GRIDGCN_EINVAL;
    if (B < 1 || n < 1 || m < 1 || k < 1 || k > 6) return GRIDGCN_EINVAL;  // best[6]
    return gg_ball_knn(unknown, known, downnum, upnum, B, n, m, k, radius, idx,
                       (hipStream_t)stream, ldu, ldk, zero_tail ? 1 : 0);
}

int gridgcn_knn(const float *unknown, const float *known, const int32_t *downnum,
                const int32_t *upnum, int B, int n, int m, int k, int32_t *idx, void *stream)
{
    if (!unknown || !known || !downnum || !upnum || !idx) return GRIDGCN_EINVAL;
    if (B < 1 || n < 1 || m < 1 || k < 1 || k > 64) return GRIDGCN_EINVAL;
    return gg_knn(unknown, known, downnum, upnum, B, n, m, k, idx, (hipStream_t)stream);
}

int gridgcn_batch_take(const float *data, const int32_t *index, int B, int N, int C, int M,
                       float *out, void *stream)
{
    if (!data || !index || !out || B < 1 || N < 1 || C < 1 || M < 1) return GRIDGCN_EINVAL;
    return gg_batch_take(data, index, B, N, C, M, out, (hipStream_t)stream);
}

int gridgcn_batch_take_backward(const float *grad_out, const int32_t *index, int B, int N, int C,
                                int M, float *grad_data, void *stream)
{
    if (!grad_out || !index || !grad_data || B < 1 || N < 1 || C < 1 || M < 1)
        return GRIDGCN_EINVAL;
    return gg_batch_take_backward(grad_out, index, B, N, C, M, grad_data, C, C,
                                  (hipStream_t)stream);
}

int gridgcn_batch_take_backward_sorted(const float *grad_out, const int32_t *index, int B, int N,
                                       int C, int M, float *grad_data, void *workspace,
                                       size_t workspace_bytes, void *stream)
{
    if (!grad_out || !index || !grad_data || B < 1 || N < 1 || C < 1 || M < 1)
        return GRIDGCN_EINVAL;
    if (!workspace || workspace_bytes < gg_take_bwd_sorted_workspace(B, N, M))
        return GRIDGCN_EWORKSPACE;
    const int rc = gg_take_bwd_sorted(grad_out, index, B, N, C, M, grad_data, C, C, workspace,
                                      (hipStream_t)stream);
    if (rc != 1) return rc;              // 1: row width not covered by the sorted path
    return gg_batch_take_backward(grad_out, index, B, N, C, M, grad_data, C, C,
                                  (hipStream_t)stream);
}

int gridgcn_linear_fwd_ld(const float *X, long long E, int cin, const float *W, const float *b, int K,
                          int ldw, int cout, const float *scale, const float *shift, float *Z,
                          double *sums, int ldz, void *stream)
{
    if (!X || !W || !b || !Z || !sums || E < 1 || cin < 1 || cout < 1 || cout > ldw ||
        (ldz && ldz < cout))
        return GRIDGCN_EINVAL;
    GGLinFwd p;
    p.X = X; p.W = W; p.b = b; p.scale = scale; p.shift = shift; p.Z = Z; p.sums = sums;
    p.E = E; p.cin = cin; p.K = K; p.ldw = ldw; p.cout = cout; p.lda = 0; p.ldz = ldz;
    int rc = gg_linear_fwd(p, (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_linear_fwd(const float *X, long long E, int cin, const float *W, const float *b, int K,
                       int ldw, int cout, const float *scale, const float *shift, float *Z,
                       double *sums, void *stream)
{
    return gridgcn_linear_fwd_ld(X, E, cin, W, b, K, ldw, cout, scale, shift, Z, sums, 0, stream);
}

int gridgcn_linear_bwd_workspace_bytes(long long E, int cin, int C, size_t *bytes)
{
    if (!bytes || E < 1 || cin < 1 || C < 1) return GRIDGCN_EINVAL;
    return gg_linear_bwd_workspace(E, cin, C, bytes, nullptr);
}

int gridgcn_linear_bwd(const float *dY, const float *Z, const float *scale, const float *shift,
                       const float *mean, const float *rstd, const float *m1, const float *m2,
                       const float *Aprev, const float *pscale, const float *pshift,
                       const float *pmean, const float *prstd, const float *Wb, const float *Wg,
                       const float *Wdx, int ndx, long long E,
                       int C, int cin, int cin_w, int rot, int ldy, float *dX, float *dW,
                       double *psums,
                       const uint8_t *amax, const float *gval, int P, void *workspace,
                       size_t workspace_bytes, void *stream)
{
    return gridgcn_linear_bwd_ld(dY, Z, scale, shift, mean, rstd, m1, m2, Aprev, pscale, pshift, pmean,
                                 prstd, Wb, Wg, Wdx, ndx, E, C, cin, cin_w, rot, ldy, 0, 0, 0, dX, dW, psums,
                                 amax, gval, P, workspace, workspace_bytes, stream);
}

static int linear_bwd_common(const double *bsums, float *dgamma, float *dbeta, float *m1w, float *m2w,
                             const float *dY, const void *Z_, const float *scale, const float *shift,
                          const float *mean, const float *rstd, const float *m1, const float *m2,
                          const float *Aprev, const float *pscale, const float *pshift,
                          const float *pmean, const float *prstd, const float *Wb, const float *Wg,
                          const float *Wdx, int ndx, long long E,
                          int C, int cin, int cin_w, int rot, int ldy, int ldz, int nbn, int zfmt,
                          float *dX, float *dW, double *psums,
                          const uint8_t *amax, const float *gval, int P, void *workspace,
                          size_t workspace_bytes, void *stream)
{
    const float *Z = (const float *)Z_;
    if ((ldz && ldz < C) || nbn < 0 || nbn > cin || (nbn & 31) || (zfmt != 0 && zfmt != 1))
        return GRIDGCN_EINVAL;
    if (amax && (!gval || P < 1 || P > 256 || E % P != 0)) return GRIDGCN_EINVAL;   // one-byte arg max
    if (Wdx && (ndx < 1 || ndx > cin)) return GRIDGCN_EINVAL;
    if (cin_w < 1 || cin_w > cin || rot < 0 || rot > cin_w) return GRIDGCN_EINVAL;
    if (!dY && amax) dY = Z;     // unused in sparse mode
    if (!dY || !Z || !scale || !shift || !mean || !rstd || !Aprev || !Wb || !dW) return GRIDGCN_EINVAL;
    if (bsums ? (!m1w || !m2w || !dgamma || !dbeta) : (!m1 || !m2)) return GRIDGCN_EINVAL;
    if (pscale && (!pshift || !pmean || !prstd || (dX && !psums))) return GRIDGCN_EINVAL;
    size_t need = 0;
    gg_linear_bwd_workspace(E, cin, C, &need, nullptr);
    if (!workspace || workspace_bytes < need) return GRIDGCN_EWORKSPACE;
    GGLinBwd p = {};
    p.dY = dY; p.Z = Z; p.scale = scale; p.shift = shift; p.mean = mean; p.rstd = rstd;
    p.m1 = m1; p.m2 = m2; p.Aprev = Aprev; p.pscale = pscale; p.pshift = pshift; p.pmean = pmean;
    p.prstd = prstd; p.Wb = Wb; p.Wg = Wg; p.dX = dX; p.dWpart = (float *)workspace; p.dW = dW;
    p.psums = psums; p.E = E; p.C = C; p.cin = cin; p.ldd = 0; p.lda = 0;
    p.amax = amax; p.gval = gval; p.P = P > 0 ? P : 1;
    p.Wdx = Wdx; p.ndx = Wdx ? ndx : cin;
    p.cin_w = cin_w; p.rot = rot;
    p.ldy = (dY && !amax && ldy > 0) ? ldy : C;
    if (p.ldy < C) return GRIDGCN_EINVAL;
    p.ldz = ldz;
    p.nbn = nbn;
    p.zfmt = zfmt;
    p.bsums = bsums; p.fin_m1 = m1w; p.fin_m2 = m2w; p.fin_dgamma = dgamma; p.fin_dbeta = dbeta;
    int rc = gg_linear_bwd(p, (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_linear_bwd_ld(const float *dY, const void *Z_, const float *scale, const float *shift,
                          const float *mean, const float *rstd, const float *m1, const float *m2,
                          const float *Aprev, const float *pscale, const float *pshift,
                          const float *pmean, const float *prstd, const float *Wb, const float *Wg,
                          const float *Wdx, int ndx, long long E,
                          int C, int cin, int cin_w, int rot, int ldy, int ldz, int nbn, int zfmt,
                          float *dX, float *dW, double *psums,
                          const uint8_t *amax, const float *gval, int P, void *workspace,
                          size_t workspace_bytes, void *stream)
{
    return linear_bwd_common(nullptr, nullptr, nullptr, nullptr, nullptr, dY, Z_, scale, shift, mean, rstd, m1,
                             m2, Aprev, pscale, pshift, pmean, prstd, Wb, Wg, Wdx, ndx, E, C, cin, cin_w, rot,
                             ldy, ldz, nbn, zfmt, dX, dW, psums, amax, gval, P, workspace, workspace_bytes,
                             stream);
}

int gridgcn_linear_bwd_fin(const float *dY, const void *Z_, const float *scale, const float *shift,
                           const float *mean, const float *rstd, const double *sums, float *m1, float *m2,
                           float *dgamma, float *dbeta,
                           const float *Aprev, const float *pscale, const float *pshift,
                           const float *pmean, const float *prstd, const float *Wb, const float *Wg,
                           const float *Wdx, int ndx, long long E,
                           int C, int cin, int cin_w, int rot, int ldy, int ldz, int nbn, int zfmt,
                           float *dX, float *dW, double *psums,
                           const uint8_t *amax, const float *gval, int P, void *workspace,
                           size_t workspace_bytes, void *stream)
{
    if (!sums) return GRIDGCN_EINVAL;
    return linear_bwd_common(sums, dgamma, dbeta, m1, m2, dY, Z_, scale, shift, mean, rstd, nullptr, nullptr,
                             Aprev, pscale, pshift, pmean, prstd, Wb, Wg, Wdx, ndx, E, C, cin, cin_w, rot,
                             ldy, ldz, nbn, zfmt, dX, dW, psums, amax, gval, P, workspace, workspace_bytes,
                             stream);
}

int gridgcn_gemm_small_workspace_bytes(int M, int N, int K, size_t *bytes)
{
    if (!bytes || M < 1 || N < 1 || K < 1) return GRIDGCN_EINVAL;
    *bytes = gg_gemm_small_workspace(M, N, K);
    return GRIDGCN_OK;
}

int gridgcn_gemm_small(int mode, const float *A, int lda, const float *B, int ldb, float *C, int ldc,
                       int M, int N, int K, int zero_left, void *workspace, size_t workspace_bytes,
                       void *stream)
{
    if (lda < 1 || ldb < 1 || ldc < 1 || M < 1 || N < 1 || K < 1) return GRIDGCN_EINVAL;
    if (mode == 2 && (!workspace || workspace_bytes < gg_gemm_small_workspace(M, N, K))) return GRIDGCN_EWORKSPACE;
    const int rc = gg_gemm_small(mode, A, lda, B, ldb, C, ldc, M, N, K, zero_left, nullptr, workspace, (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_gemm_bias(int mode, const float *A, int lda, const float *B, int ldb, const float *bias, float *C,
                      int ldc, int M, int N, int K, void *stream)
{
    if (lda < 1 || ldb < 1 || ldc < 1 || M < 1 || N < 1 || K < 1 || (mode != 0 && mode != 1)) return GRIDGCN_EINVAL;
    const int rc = gg_gemm_small(mode, A, lda, B, ldb, C, ldc, M, N, K, 0, bias, nullptr, (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_linear_dx(const float *dY, const float *Z, const float *scale, const float *shift,
                      const float *mean, const float *rstd, const float *m1, const float *m2,
                      const float *Aprev, const float *pscale, const float *pshift,
                      const float *pmean, const float *prstd, const float *Wdx, int ndx,
                      long long E, int C, int cin, int ldy, float drop_p, uint64_t drop_seed,
                      const uint64_t *drop_seed_dev, float *dX, double *psums, void *stream)
{
    if (!dY || !Z || !scale || !shift || !mean || !rstd || !m1 || !m2 || !Wdx || !dX)
        return GRIDGCN_EINVAL;
    if (E < 1 || C < 1 || cin < 1 || ndx < 1 || ndx > cin || ldy < C) return GRIDGCN_EINVAL;
    if (pscale && (!pshift || !pmean || !prstd || !psums || !Aprev)) return GRIDGCN_EINVAL;
    if (!(drop_p >= 0.f && drop_p < 1.f)) return GRIDGCN_EINVAL;
    GGLinBwd p = {};
    p.dY = dY; p.Z = Z; p.scale = scale; p.shift = shift; p.mean = mean; p.rstd = rstd;
    p.m1 = m1; p.m2 = m2; p.Aprev = Aprev ? Aprev : dX; p.pscale = pscale; p.pshift = pshift;
    p.pmean = pmean; p.prstd = prstd; p.dX = dX; p.psums = psums; p.E = E; p.C = C; p.cin = cin;
    p.P = 1; p.Wdx = Wdx; p.ndx = ndx; p.cin_w = cin; p.ldy = ldy;
    gg_drop_consts(drop_p, &p.drop_thr, &p.drop_scale);
    p.drop_lo = (unsigned)drop_seed; p.drop_hi = (unsigned)(drop_seed >> 32);
    p.drop_dev = (const unsigned long long *)drop_seed_dev;
    const int rc = gg_linear_dx_direct(p, (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_bn_relu_dropout_apply(const float *Z, const float *scale, const float *shift, float *Y,
                                  long long E, int C, int ldy, float drop_p, uint64_t drop_seed,
                                  const uint64_t *drop_seed_dev, void *stream)
{
    if (!Z || !scale || !shift || !Y || E < 1 || C < 1 || ldy < C) return GRIDGCN_EINVAL;
    if (!(drop_p >= 0.f && drop_p < 1.f)) return GRIDGCN_EINVAL;
    return gg_bn_apply(Z, scale, shift, Y, E, C, ldy, drop_p, drop_seed,
                       (const unsigned long long *)drop_seed_dev, (hipStream_t)stream);
}

int gridgcn_pairmax_fwd(const float *Zp, const float *Za, const float *scale_p,
                        const float *shift_p, const float *scale_a, const float *shift_a,
                        long long ncent, int P, int C, float *agg, int ld_agg, uint8_t *amax,
                        float *zsel, void *stream)
{
    if (P > 256) return GRIDGCN_EINVAL;   // one-byte arg max
    if (!Zp || !Za || !scale_p || !shift_p || !scale_a || !shift_a || !agg || !amax || ncent < 1 ||
        P < 1 || C < 1 || ld_agg < C)
        return GRIDGCN_EINVAL;
    return gg_pairmax_fwd(Zp, Za, scale_p, shift_p, scale_a, shift_a, ncent, P, C, agg, ld_agg,
                          amax, zsel, (hipStream_t)stream);
}

int gridgcn_pairmax_fwd_src_z(const float *Ysrc, const int32_t *nebidx, const float *att16,
                              const float *Wg, const float *b, int B, int Nsrc, int O,
                              const void *Za, int za_bf16, const float *scale_p, const float *shift_p,
                              const float *scale_a, const float *shift_a, long long ncent, int P,
                              int C, float *agg, int ld_agg, uint8_t *amax, float *zsel,
                              void *stream)
{
    if ((!Ysrc && !Wg) || !nebidx || !att16 || !b || !Za || !scale_p || !shift_p || !scale_a ||
        !shift_a || !agg || !amax || ncent < 1 || P < 1 || P > 256 || C < 1 || B < 1 || Nsrc < 1 || O < 1 ||
        ncent != (long long)B * O || ld_agg < C)
        return GRIDGCN_EINVAL;
    const int rc = gg_pairmax_fwd_src(Ysrc, nebidx, att16, Wg, b, B, Nsrc, O, (const float *)Za, scale_p,
                                      shift_p, scale_a, shift_a, ncent, P, C, agg, ld_agg, amax, zsel,
                                      za_bf16 ? 1 : 0, (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_pairmax_fwd_src(const float *Ysrc, const int32_t *nebidx, const float *att16,
                            const float *Wg, const float *b, int B, int Nsrc, int O,
                            const float *Za, const float *scale_p, const float *shift_p,
                            const float *scale_a, const float *shift_a, long long ncent, int P,
                            int C, float *agg, int ld_agg, uint8_t *amax, float *zsel,
                            void *stream)
{
    return gridgcn_pairmax_fwd_src_z(Ysrc, nebidx, att16, Wg, b, B, Nsrc, O, Za, 0, scale_p, shift_p,
                                     scale_a, shift_a, ncent, P, C, agg, ld_agg, amax, zsel, stream);
}

int gridgcn_pairmax_bwd(const float *Zp, const float *Za, const float *scale_p,
                        const float *shift_p, const float *mean_p, const float *rstd_p,
                        const float *scale_a, const float *shift_a, const float *mean_a,
                        const float *rstd_a, const float *dagg, const uint8_t *amax,
                        long long ncent, int P, int C, int ld_dagg, float *gp, float *ga,
                        double *sums_p, double *sums_a, const float *zsel, void *stream)
{
    if (((!Zp || !Za) && !zsel) || !dagg || !amax || !gp || !ga || !sums_p || !sums_a ||
        ncent < 1 || P < 1 || P > 256 || ld_dagg < C)
        return GRIDGCN_EINVAL;
    int rc = gg_pairmax_bwd(Zp, Za, scale_p, shift_p, mean_p, rstd_p, scale_a, shift_a, mean_a,
                            rstd_a, dagg, amax, ncent, P, C, ld_dagg, gp, ga, sums_p, sums_a, zsel, 0,
                            (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_pairmax_bwd_masked(const float *scale_p, const float *shift_p, const float *mean_p,
                               const float *rstd_p, const float *scale_a, const float *shift_a,
                               const float *mean_a, const float *rstd_a, const float *dagg,
                               const uint8_t *amax, long long ncent, int P, int C, int ld_dagg, float *gp,
                               float *ga, double *sums_p, double *sums_a, const float *zsel, void *stream)
{
    if (!zsel || !dagg || !amax || !gp || !ga || !sums_p || !sums_a || ncent < 1 || P < 1 || P > 256 ||
        ld_dagg < C)
        return GRIDGCN_EINVAL;
    int rc = gg_pairmax_bwd(nullptr, nullptr, scale_p, shift_p, mean_p, rstd_p, scale_a, shift_a, mean_a,
                            rstd_a, dagg, amax, ncent, P, C, ld_dagg, gp, ga, sums_p, sums_a, zsel, 1,
                            (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_att_bwd_noz_workspace_bytes(long long E, int cin, int C, size_t *bytes)
{
    if (!bytes || E < 1) return GRIDGCN_EINVAL;
    if (!gg_att_bwd_noz_ok(E, cin, C)) return GRIDGCN_EINVAL;
    *bytes = gg_att_bwd_noz_workspace(E);
    return GRIDGCN_OK;
}

int gridgcn_att_bwd_noz(const float *Z1, const float *pscale, const float *pshift, const float *pmean,
                        const float *prstd, const float *W2, const float *b2, const float *scale,
                        const float *mean, const float *rstd, const double *sums, const uint8_t *amax,
                        const float *gval, int P, long long E, int cin, int C, float *dX, float *dW, float *m1,
                        float *m2, float *dgamma, float *dbeta, double *psums, double *s1, void *workspace,
                        size_t workspace_bytes, void *stream)
{
    if (!Z1 || !pscale || !pshift || !pmean || !prstd || !W2 || !b2 || !scale || !mean || !rstd || !sums ||
        !amax || !gval || !dX || !dW || !m1 || !m2 || !dgamma || !dbeta || !psums || !s1)
        return GRIDGCN_EINVAL;
    if (!gg_att_bwd_noz_ok(E, cin, C) || P < 1 || P > 256 || (E % P)) return GRIDGCN_EINVAL;
    if (!workspace || workspace_bytes < gg_att_bwd_noz_workspace(E)) return GRIDGCN_EWORKSPACE;
    const int rc = gg_att_bwd_noz(Z1, pscale, pshift, pmean, prstd, W2, b2, scale, mean, rstd, sums, amax, gval,
                                  P, E, dX, dW, m1, m2, dgamma, dbeta, psums, s1, workspace, (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_att_bwd_noz_mom_supported(long long E, int cin, int C, int P)
{
    return gg_att_bwd_noz_mom_ok(E, cin, C, P) ? 1 : 0;
}

int gridgcn_att_bwd_noz_mom(const float *Z1, const float *pscale, const float *pshift, const float *pmean,
                            const float *prstd, const float *W2, const float *b2, const float *scale,
                            const float *mean, const float *rstd, const double *sums, const uint8_t *amax,
                            const float *gval, int P, long long E, int cin, int C, const double *moments, float *dX,
                            float *dW, float *m1, float *m2, float *dgamma, float *dbeta, double *psums,
                            void *workspace, size_t workspace_bytes, void *stream)
{
    if (!Z1 || !pscale || !pshift || !pmean || !prstd || !W2 || !b2 || !scale || !mean || !rstd || !sums ||
        !amax || !gval || !moments || !dX || !dW || !m1 || !m2 || !dgamma || !dbeta || !psums)
        return GRIDGCN_EINVAL;
    if (P < 1 || P > 256 || !gg_att_bwd_noz_mom_ok(E, cin, C, P)) return GRIDGCN_EINVAL;
    if (!workspace || workspace_bytes < gg_att_bwd_noz_workspace(E)) return GRIDGCN_EWORKSPACE;
    const int rc = gg_att_bwd_noz(Z1, pscale, pshift, pmean, prstd, W2, b2, scale, mean, rstd, sums, amax, gval,
                                  P, E, dX, dW, m1, m2, dgamma, dbeta, psums, nullptr, workspace, (hipStream_t)stream,
                                  moments);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_att_moments_offset(long long E, int cin, int C, size_t *offset_bytes)
{
    if (!offset_bytes || E < 1 || E >= (1ll << 25) || cin != 32 || C != 128) return GRIDGCN_EINVAL;
    *offset_bytes = (size_t)gg_att_moments_grid_of(E) * 17 * 64 * sizeof(double);
    return GRIDGCN_OK;
}

int gridgcn_att_fwd_noz_workspace_bytes(long long E, int cin, int C, size_t *bytes)
{
    if (!bytes || E < 1 || E >= (1ll << 25) || cin != 32 || C != 128) return GRIDGCN_EINVAL;
    *bytes = gg_att_moments_workspace(E);
    return GRIDGCN_OK;
}

int gridgcn_att_pairmax_fwd_supported(long long ncent, int O, int P, int cin, int C, int ld_agg, long long rows)
{
    return gg_att_fwd_ok(ncent, O, P, cin, C, ld_agg, rows) ? 1 : 0;
}

int gridgcn_att_bn2_moments(const float *Z1, const float *scale1, const float *shift1, const float *W2,
                            const float *b2, const float *gamma, const float *beta, long long E, int cin, int C,
                            float eps, float momentum, float *scale, float *shift, float *mean, float *rstd,
                            float *running_mean, float *running_var, int64_t *num_batches_tracked, double *sums,
                            void *workspace, size_t workspace_bytes, void *stream)
{
    if (!Z1 || !scale1 || !shift1 || !W2 || !b2 || !gamma || !beta || !scale || !shift || !mean || !rstd ||
        E < 1 || E >= (1ll << 25) || cin != 32 || C != 128 || (!running_mean) != (!running_var))
        return GRIDGCN_EINVAL;
    if (!workspace || workspace_bytes < gg_att_moments_workspace(E)) return GRIDGCN_EWORKSPACE;
    return gg_att_bn2_moments(Z1, scale1, shift1, W2, b2, gamma, beta, E, eps, momentum, scale, shift, mean, rstd,
                              running_mean, running_var, (long long *)num_batches_tracked, sums, workspace,
                              (hipStream_t)stream);
}

int gridgcn_att_pairmax_fwd(const float *Ysrc, const int32_t *nebidx, const float *att16, const float *Wg,
                            const float *b, int B, int Nsrc, int O, const float *Z1, const float *scale1,
                            const float *shift1, const float *W2, const float *b2, const float *scale_p,
                            const float *shift_p, const float *scale_a, const float *shift_a, long long ncent,
                            int P, int cin, int C, float *agg, int ld_agg, uint8_t *amax, float *zsel, void *stream)
{
    if (!Ysrc || !nebidx || !att16 || !b || !Z1 || !scale1 || !shift1 || !W2 || !b2 || !scale_p || !shift_p ||
        !scale_a || !shift_a || !agg || !amax || !zsel || B < 1 || Nsrc < 1 || O < 1 || ncent != (long long)B * O)
        return GRIDGCN_EINVAL;
    if (!gg_att_fwd_ok(ncent, O, P, cin, C, ld_agg, (long long)B * Nsrc)) return GRIDGCN_EINVAL;
    return gg_att_pairmax_args(Ysrc, nebidx, att16, Wg, b, B, Nsrc, O, Z1, scale1, shift1, W2, b2, scale_p, shift_p,
                               scale_a, shift_a, ncent, agg, ld_agg, amax, zsel, (hipStream_t)stream);
}

int gridgcn_bn_relu_apply(const float *Z, const float *scale, const float *shift, float *Y,
                          long long E, int C, int ldy, void *stream)
{
    if (!Z || !scale || !shift || !Y || E < 1 || C < 1 || ldy < C) return GRIDGCN_EINVAL;
    return gg_bn_apply(Z, scale, shift, Y, E, C, ldy, 0.f, 0ull, nullptr, (hipStream_t)stream);
}

int gridgcn_bn_relu_bwd_reduce(const float *dY, const float *Z, const float *scale,
                               const float *shift, const float *mean, const float *rstd,
                               long long E, int C, int ldy, double *sums, void *stream)
{
    if (!dY || !Z || !scale || !shift || !mean || !rstd || !sums || E < 1 || C < 1 || ldy < C)
        return GRIDGCN_EINVAL;
    int rc = gg_bn_bwd_reduce(dY, Z, scale, shift, mean, rstd, E, C, sums, ldy,
                              (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_bn_relu_bwd_elemt(const float *dY, const float *Z, const float *scale,
                              const float *shift, const float *mean, const float *rstd,
                              const float *m1, const float *m2, long long E, int C, float *dZ,
                              void *stream)
{
    if (!dY || !Z || !scale || !shift || !mean || !rstd || !m1 || !m2 || !dZ || E < 1 || C < 1)
        return GRIDGCN_EINVAL;
    return gg_bn_bwd_elemt(dY, Z, scale, shift, mean, rstd, m1, m2, E, C, dZ, (hipStream_t)stream);
}

int gridgcn_pack_desc_fill(gridgcn_pack_desc *desc_host)
{
    return gg_pack_desc_fill(desc_host) ? GRIDGCN_EINVAL : GRIDGCN_OK;
}

int gridgcn_pack_linear_batch(const gridgcn_pack_desc *descs_dev, int nlayers, int max_n, void *stream)
{
    const int rc = gg_pack_linear_batch(descs_dev, nlayers, max_n, (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_pack_linear(const float *W, const float *b, int C, int cin_w, int rot, int cin, int ndx,
                        float *Wp, float *Bp, float *Wb, float *Wg, float *Wq, float *Wdx,
                        void *stream)
{
    if (!W || (Bp && !b)) return GRIDGCN_EINVAL;
    int rc = gg_pack_linear(W, b, C, cin_w, rot, cin, ndx, Wp, Bp, Wb, Wg, Wq, Wdx,
                            (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_linear_fwd_direct_ld(const float *X, long long E, int K, int ldx, const float *Wq,
                                 const float *b, int ldw, int cout, const float *scale,
                                 const float *shift, void *Z, double *sums, int ldz, int zfmt,
                                 void *stream)
{
    if (!X || !Wq || !b || (!Z && !sums) || cout < 1 || cout > ldw || (scale && !shift) || ldx < K ||
        (ldx & 3) || ((uintptr_t)X & 15) || (ldz && ldz < cout) || (zfmt != 0 && zfmt != 1))
        return GRIDGCN_EINVAL;
    GGLinFwd p;
    p.X = X; p.W = Wq; p.b = b; p.scale = scale; p.shift = shift; p.Z = (float *)Z; p.sums = sums;
    p.E = E; p.cin = K; p.K = K; p.ldw = ldw; p.cout = cout; p.lda = ldx; p.ldz = ldz; p.zfmt = zfmt;
    int rc = gg_linear_fwd_direct(p, (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_linear_fwd_direct_fin(const float *X, long long E, int K, int ldx, const float *Wq,
                                  const float *b, int ldw, int cout, const float *scale,
                                  const float *shift, void *Z, double *sums, int ldz, int zfmt,
                                  const gridgcn_bn_fin *fin, void *stream)
{
    if (!X || !Wq || !b || !sums || cout < 1 || cout > ldw || (scale && !shift) || ldx < K ||
        (ldx & 3) || ((uintptr_t)X & 15) || (ldz && ldz < cout) || (zfmt != 0 && zfmt != 1))
        return GRIDGCN_EINVAL;
    if (!fin || !fin->gamma || !fin->beta || !fin->scale || !fin->shift || !fin->mean || !fin->rstd ||
        !fin->ticket || fin->tail < 0 || (fin->running_mean && !fin->running_var) || E < 1)
        return GRIDGCN_EINVAL;
    GGLinFwd p;
    p.X = X; p.W = Wq; p.b = b; p.scale = scale; p.shift = shift; p.Z = (float *)Z; p.sums = sums;
    p.E = E; p.cin = K; p.K = K; p.ldw = ldw; p.cout = cout; p.lda = ldx; p.ldz = ldz; p.zfmt = zfmt;
    p.gamma = fin->gamma; p.beta = fin->beta; p.fscale = fin->scale; p.fshift = fin->shift;
    p.fmean = fin->mean; p.frstd = fin->rstd; p.run_mean = fin->running_mean; p.run_var = fin->running_var;
    p.nbt = (long long *)fin->num_batches_tracked; p.fin_ticket = fin->ticket; p.eps = fin->eps;
    p.momentum = fin->momentum; p.fin_tail = fin->tail;
    int rc = gg_linear_fwd_direct(p, (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_linear_fwd_direct_drop(const float *X, long long E, int K, int ldx, const float *Wq,
                                   const float *b, int ldw, int cout, const float *scale,
                                   const float *shift, float *Z, float drop_p, uint64_t drop_seed,
                                   const uint64_t *drop_seed_dev, void *stream)
{
    if (!X || !Wq || !b || !Z || !scale || !shift || cout < 1 || cout > 32 || cout > ldw || ldx < K ||
        (ldx & 3) || ((uintptr_t)X & 15) || (K & 31) || E < 1 || !(drop_p > 0.f && drop_p < 1.f))
        return GRIDGCN_EINVAL;
    GGLinFwd p;
    p.X = X; p.W = Wq; p.b = b; p.scale = scale; p.shift = shift; p.Z = Z; p.sums = nullptr;
    p.E = E; p.cin = K; p.K = K; p.ldw = ldw; p.cout = cout; p.lda = ldx;
    gg_drop_consts(drop_p, &p.drop_thr, &p.drop_scale);
    if (!p.drop_thr) return GRIDGCN_EINVAL;
    p.drop_lo = (unsigned)drop_seed; p.drop_hi = (unsigned)(drop_seed >> 32);
    p.drop_dev = (const unsigned long long *)drop_seed_dev;
    int rc = gg_linear_fwd_direct(p, (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_linear_dw_drop(const float *dY, const float *Z, const float *scale, const float *shift,
                           const float *mean, const float *rstd, const float *m1, const float *m2,
                           const float *Aprev, const float *pscale, const float *pshift, long long E, int C,
                           int cin, float drop_p, uint64_t drop_seed, const uint64_t *drop_seed_dev,
                           float *dW, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!dY || !Z || !scale || !shift || !mean || !rstd || !m1 || !m2 || !Aprev || !pscale || !pshift || !dW ||
        E < 1 || C < 1 || cin < 1 || !(drop_p > 0.f && drop_p < 1.f))
        return GRIDGCN_EINVAL;
    const size_t need = gg_linear_dw_direct_workspace(E, cin, C);
    if (!need) return GRIDGCN_EINVAL;
    if (!workspace || workspace_bytes < need) return GRIDGCN_EWORKSPACE;
    GGLinBwd p = {};
    p.dY = dY; p.Z = Z; p.scale = scale; p.shift = shift; p.mean = mean; p.rstd = rstd; p.m1 = m1; p.m2 = m2;
    p.Aprev = Aprev; p.pscale = pscale; p.pshift = pshift; p.pmean = pscale; p.prstd = pscale;   // (unused by dW)
    p.dWpart = (float *)workspace; p.dW = dW; p.E = E; p.C = C; p.cin = cin; p.cin_w = cin; p.ldy = C; p.P = 1;
    gg_drop_consts(drop_p, &p.drop_thr, &p.drop_scale);
    if (!p.drop_thr) return GRIDGCN_EINVAL;
    p.drop_lo = (unsigned)drop_seed; p.drop_hi = (unsigned)(drop_seed >> 32);
    p.drop_dev = (const unsigned long long *)drop_seed_dev;
    const int rc = gg_linear_dw_direct(p, (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_linear_fwd_direct(const float *X, long long E, int K, int ldx, const float *Wq,
                              const float *b, int ldw, int cout, const float *scale,
                              const float *shift, float *Z, double *sums, void *stream)
{
    return gridgcn_linear_fwd_direct_ld(X, E, K, ldx, Wq, b, ldw, cout, scale, shift, Z, sums, 0, 0,
                                        stream);
}

int gridgcn_linear_fwd_direct2(const float *X1, int ld1, int K1, const float *X2, int ld2, int K2,
                               long long E, const float *Wq, const float *b, const float *rowbias,
                               int P, int ldw, int cout, const float *scale, const float *shift,
                               float *Z, double *sums, void *stream)
{
    if (!X1 || !X2 || !Wq || !Z || (!b && !rowbias) || cout < 1 || cout > ldw ||
        (scale && !shift) || K1 < 32 || (K1 & 31) || K2 < 8 || (K2 & 7) || ld1 < K1 || ld2 < K2 ||
        ((ld1 | ld2) & 3) || (((uintptr_t)X1 | (uintptr_t)X2) & 15))
        return GRIDGCN_EINVAL;
    if (rowbias && (P < 32 || (P & 31) || E % P)) return GRIDGCN_EINVAL;
    GGLinFwd p;
    p.X = X1; p.W = Wq; p.b = b ? b : rowbias; p.scale = scale; p.shift = shift; p.Z = Z;
    p.sums = sums; p.E = E; p.cin = K1 + K2; p.K = K1 + K2; p.ldw = ldw; p.cout = cout;
    p.lda = ld1;
    p.X2 = X2; p.K1 = K1; p.lda2 = ld2; p.rowbias = rowbias; p.P = P;
    int rc = gg_linear_fwd_direct(p, (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_ctx_max(const float *src, const int32_t *nebidx, const float *cent, int cent_stride,
                    int B, int Nsrc, int Cs, int O, int P, float *ctx, int32_t *cidx, void *stream)
{
    if (!src || !nebidx || !cent || !ctx || B < 1 || Nsrc < 1 || O < 1 || P < 1 || cent_stride < 3)
        return GRIDGCN_EINVAL;
    int rc = gg_ctx_max(src, nebidx, cent, cent_stride, B, Nsrc, Cs, O, P, ctx, cidx,
                        (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_ctx_max_backward(const float *dctx, const int32_t *cidx, long long ncent, int Cf, int Cs,
                             float *dsrc, void *stream)
{
    if (!dctx || !cidx || !dsrc || ncent < 1 || Cf < 0 || Cs != Cf + 4) return GRIDGCN_EINVAL;
    return gg_ctx_scatter(dctx, cidx, ncent, Cf, Cs, dsrc, (hipStream_t)stream);
}

int gridgcn_bn_dz_segsum(const float *dY, const float *Z, const float *scale, const float *shift,
                         const float *mean, const float *rstd, const float *m1, const float *m2,
                         long long ncent, int P, int C, float *out, void *stream)
{
    if (!dY || !Z || !scale || !shift || !mean || !rstd || !m1 || !m2 || !out || ncent < 1 || P < 1)
        return GRIDGCN_EINVAL;
    int rc = gg_dz_segsum(dY, Z, scale, shift, mean, rstd, m1, m2, ncent, P, C, out,
                          (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_bn_stats(const float *Z, long long E, int C, int ld, double *sums, void *stream)
{
    if (!Z || !sums || E < 1 || C < 1 || ld < C) return GRIDGCN_EINVAL;
    return gg_bn_stats(Z, E, C, ld, sums, (hipStream_t)stream);
}

int gridgcn_sparse_add(const uint8_t *amax, const float *gval, long long ncent, int P, int C,
                       float *dX, void *stream)
{
    if (!amax || !gval || !dX || ncent < 1 || P < 1 || P > 256 || C < 1) return GRIDGCN_EINVAL;
    return gg_sparse_add(amax, gval, ncent, P, C, dX, (hipStream_t)stream);
}

int gridgcn_bn_finalize(const double *sums, const float *gamma, const float *beta, long long E,
                        float eps, float momentum, int C, float *scale, float *shift, float *mean,
                        float *rstd, float *running_mean, float *running_var,
                        int64_t *num_batches_tracked, void *stream)
{
    if (!sums || !gamma || !beta || !scale || !shift || !mean || !rstd || E < 1 || C < 1 ||
        (running_mean && !running_var))
        return GRIDGCN_EINVAL;
    return gg_bn_finalize(sums, gamma, beta, E, eps, momentum, C, scale, shift, mean, rstd,
                          running_mean, running_var, (long long *)num_batches_tracked,
                          (hipStream_t)stream);
}

int gridgcn_bn_finalize_tail(const double *sums, const float *gamma, const float *beta, long long E,
                             float eps, float momentum, int C, int tail, float *scale, float *shift,
                             float *mean, float *rstd, float *running_mean, float *running_var,
                             int64_t *num_batches_tracked, void *stream)
{
    if (!sums || !gamma || !beta || !scale || !shift || !mean || !rstd || E < 1 || C < 1 || tail < 0 ||
        (running_mean && !running_var))
        return GRIDGCN_EINVAL;
    return gg_bn_finalize(sums, gamma, beta, E, eps, momentum, C, scale, shift, mean, rstd,
                          running_mean, running_var, (long long *)num_batches_tracked,
                          (hipStream_t)stream, tail);
}

int gridgcn_bn_bwd_finalize(const double *sums, long long E, int C, float *m1, float *m2,
                            float *dgamma, float *dbeta, void *stream)
{
    if (!sums || !m1 || !m2 || !dgamma || !dbeta || E < 1 || C < 1) return GRIDGCN_EINVAL;
    return gg_bn_bwd_finalize(sums, E, C, m1, m2, dgamma, dbeta, (hipStream_t)stream);
}

int gridgcn_edge_inputs(const float *src, const int32_t *nebidx, const float *cent,
                        int cent_stride, int B, int Nsrc, int Cs, int O, int P, int has_feats,
                        int localfdim, float *nf, float *att, void *stream)
{
    if (!src || !nebidx || !cent || !nf || !att) return GRIDGCN_EINVAL;
    if (B < 1 || Nsrc < 1 || O < 1 || P < 1 || Cs < 4 || (has_feats && Cs == 4))
        return GRIDGCN_EINVAL;
    return gg_edge_inputs(src, nebidx, cent, cent_stride, B, Nsrc, Cs, O, P, has_feats, localfdim,
                          nf, att, (hipStream_t)stream);
}

int gridgcn_edge_inputs_backward(const float *grad_nf, const int32_t *nebidx, int B, int Nsrc,
                                 int Cs, int O, int P, int has_feats, int localfdim,
                                 float *grad_src, void *stream)
{
    if (!grad_nf || !nebidx || !grad_src || B < 1 || Nsrc < 1 || O < 1 || P < 1 || Cs <= 4 ||
        !has_feats)
        return GRIDGCN_EINVAL;
    const int fo = localfdim != 0 ? 3 : 0;
    const int cin = fo + Cs - 4;
    // only the feature columns carry gradient: xyz/w come from the non-differentiable index ops
    return gg_batch_take_backward(grad_nf + fo, nebidx, B, Nsrc, Cs - 4, O * P, grad_src + 4, cin,
                                  Cs, (hipStream_t)stream);
}

int gridgcn_softmax_ce_fwd(const float *logits, int ld, int ncls, const int64_t *label, long long E,
                           int ignore_label, float *lse, double *acc, void *stream)
{
    if (!logits || !label || !lse || !acc) return GRIDGCN_EINVAL;
    int rc = gg_ce_fwd(logits, ld, ncls, (const long long *)label, E, ignore_label, lse, acc, nullptr,
                       (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_softmax_ce_loss(const float *logits, int ld, int ncls, const int64_t *label, long long E,
                            int ignore_label, float *lse, double *acc3, float *loss, void *stream)
{
    if (!logits || !label || !lse || !acc3 || !loss) return GRIDGCN_EINVAL;
    int rc = gg_ce_fwd(logits, ld, ncls, (const long long *)label, E, ignore_label, lse, acc3, loss,
                       (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_softmax_ce_bwd(const float *logits, int ld, int ncls, const int64_t *label, long long E,
                           int ignore_label, const float *lse, const double *acc,
                           const float *grad_loss, const float *class_weight, float *dlogits,
                           void *stream)
{
    if (!logits || !label || !lse || !acc || !grad_loss || !dlogits) return GRIDGCN_EINVAL;
    int rc = gg_ce_bwd(logits, ld, ncls, (const long long *)label, E, ignore_label, lse, acc,
                       grad_loss, class_weight, dlogits, (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_colsum(const float *X, long long E, int ld, int ncols, double *out, void *stream)
{
    if (!X || !out) return GRIDGCN_EINVAL;
    int rc = gg_colsum(X, E, ld, ncols, out, nullptr, (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_colsum_f32(const float *X, long long E, int ld, int ncols, double *acc, float *out,
                       void *stream)
{
    if (!X || !acc || !out) return GRIDGCN_EINVAL;
    int rc = gg_colsum(X, E, ld, ncols, acc, out, (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_edge_inputs_rows(const float *src, const int32_t *nebidx, const float *cent,
                             int cent_stride, int B, int Nsrc, int Cs, int O, int P, int has_feats,
                             int localfdim, int nf_stride, float *nf, float *att16, void *stream)
{
    if (!src || !nebidx || !cent || !nf || !att16) return GRIDGCN_EINVAL;
    if (B < 1 || Nsrc < 1 || O < 1 || P < 1 || Cs < 4 || (has_feats && Cs == 4))
        return GRIDGCN_EINVAL;
    int rc = gg_edge_inputs_rows(src, nebidx, cent, cent_stride, B, Nsrc, Cs, O, P, has_feats,
                                 localfdim, nf_stride, nf, att16, (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_edge_lin0_forward(const float *Ysrc, const float *src, const int32_t *nebidx,
                              const float *cent, int cent_stride, int B, int Nsrc, int Cs, int O,
                              int P, int C0, const float *Wg, const float *b, float *Z0,
                              float *att16, double *sums, void *stream)
{
    if (!src || !nebidx || !cent || !b || !att16 || B < 1 || Nsrc < 1 || Cs < 3 ||
        O < 1 || P < 1 || C0 < 1 || (!Ysrc && !Wg) || (long long)B * O * P >= (1ll << 31))
        return GRIDGCN_EINVAL;
    GGEdgeLin0 p;
    p.Ysrc = Ysrc; p.src = src; p.nebidx = nebidx; p.cent = cent; p.Wg = Wg; p.b = b; p.Z = Z0;
    p.att16 = att16; p.sums = sums; p.cent_stride = cent_stride; p.B = B; p.Nsrc = Nsrc; p.Cs = Cs;
    p.O = O; p.P = P; p.C0 = C0; p.E = B * O * P;
    const int rc = gg_edge_lin0_fwd(p, (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_edge_geo_forward_workspace_bytes(int B, int Nsrc, int O, int P, size_t *bytes)
{
    if (!bytes || B < 1 || Nsrc < 1 || O < 1 || P < 1) return GRIDGCN_EINVAL;
    *bytes = gg_edge_geo_workspace(B, Nsrc, (long long)O * P);
    return GRIDGCN_OK;
}

int gridgcn_edge_geo_forward(const float *Ysrc, const float *src, const int32_t *nebidx, const float *cent,
                             int cent_stride, int B, int Nsrc, int Cs, int O, int P, int C0,
                             const float *Wg, const float *b, float *att16, float *Gsum, double *gg,
                             double *sums, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!Ysrc || !src || !nebidx || !cent || !b || !att16 || !Gsum || !gg || !sums || B < 1 || Nsrc < 1 ||
        Cs < 3 || O < 1 || P < 1 || C0 < 1 || (long long)B * O * P >= (1ll << 31))
        return GRIDGCN_EINVAL;
    if (!workspace || workspace_bytes < gg_edge_geo_workspace(B, Nsrc, (long long)O * P)) return GRIDGCN_EWORKSPACE;
    const int rc = gg_edge_geo_fwd(Ysrc, src, nebidx, cent, cent_stride, B, Nsrc, Cs, O, P, C0, Wg, b, att16,
                                   Gsum, gg, sums, workspace, (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_edge_lin0_backward(const float *Z0, const float *Ysrc, const float *Wg, const float *b,
                               const float *dY, const uint8_t *amax,
                               const float *gval, const float *scale, const float *shift,
                               const float *mean, const float *rstd, const float *m1,
                               const float *m2, const float *att16, const int32_t *nebidx, int B,
                               int Nsrc, int O, int P, int C0, float *dYsrc, double *dWg,
                               void *workspace, size_t workspace_bytes, void *stream)
{
    if ((!Z0 && (!b || (!Ysrc && !Wg))) || (!dY && (!amax || !gval)) || !scale || !shift || !mean ||
        !rstd || !m1 || !m2 || !att16 || !nebidx || !dYsrc || B < 1 || Nsrc < 1 || O < 1 || P < 1 ||
        (amax && P > 256) || C0 < 1)
        return GRIDGCN_EINVAL;
    if (!workspace || workspace_bytes < gg_csr_workspace(B, Nsrc, O * P)) return GRIDGCN_EWORKSPACE;
    GGEdgeLin0Bwd p;
    p.Z = Z0; p.Ysrc = Ysrc; p.Wg = Wg; p.b = b; p.dY = dY; p.amax = amax; p.gval = gval; p.scale = scale; p.shift = shift;
    p.mean = mean; p.rstd = rstd; p.m1 = m1; p.m2 = m2; p.att16 = att16; p.index = nebidx;
    p.perm = nullptr; p.keys = nullptr; p.rowptr = nullptr; p.dYsrc = dYsrc; p.dWg = dWg;
    p.B = B; p.N = Nsrc; p.O = O; p.P = P; p.C0 = C0; p.M = O * P; p.cpc = 0; p.chunk = 0;
    const int rc = gg_edge_lin0_bwd(p, workspace, (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_edge_lin0_backward_sparse_workspace_bytes(int B, int Nsrc, int C0, size_t *bytes)
{
    if (!bytes || B < 1 || Nsrc < 1 || C0 < 1) return GRIDGCN_EINVAL;
    *bytes = gg_edge_lin0_sparse_workspace(B, Nsrc, C0);
    return GRIDGCN_OK;
}

int gridgcn_edge_lin0_backward_sparse(const int32_t *nebidx, const float *att16,
                                      const uint8_t *amax, const float *gval, const float *zsel,
                                      const float *Ysrc, const float *Wg, const float *b,
                                      const float *scale, const float *shift, const float *mean,
                                      const float *rstd, const float *m1, const float *m2, int B,
                                      int Nsrc, int O, int P, int C0, float *dYsrc, float *Gsum,
                                      double *wgs, double *gg, void *workspace,
                                      size_t workspace_bytes, void *stream)
{
    if (!nebidx || !att16 || !amax || !gval || !zsel || (!Ysrc && !Wg) || !b || !scale || !shift ||
        !mean || !rstd || !m1 || !m2 || !dYsrc || !Gsum || !wgs || !gg || B < 1 || Nsrc < 1 ||
        O < 1 || P < 1 || P > 256 || C0 < 1)
        return GRIDGCN_EINVAL;
    if (!workspace || workspace_bytes < gg_edge_lin0_sparse_workspace(B, Nsrc, C0))
        return GRIDGCN_EWORKSPACE;
    const int rc = gg_edge_lin0_bwd_sparse(nebidx, att16, amax, gval, zsel, Ysrc, Wg, b, scale,
                                           shift, mean, rstd, m1, m2, B, Nsrc, O, P, C0, dYsrc, Gsum,
                                           wgs, gg, workspace, (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_edge_lin0_backward_sparse_geo(const int32_t *nebidx, const float *att16,
                                          const uint8_t *amax, const float *gval, const float *zsel,
                                          const float *Ysrc, const float *Wg, const float *b,
                                          const float *scale, const float *shift, const float *mean,
                                          const float *rstd, const float *m1, const float *m2, int B,
                                          int Nsrc, int O, int P, int C0, float *dYsrc,
                                          const float *Gsum, double *wgs, void *workspace,
                                          size_t workspace_bytes, void *stream)
{
    if (!nebidx || !att16 || !amax || !gval || !zsel || (!Ysrc && !Wg) || !b || !scale || !shift ||
        !mean || !rstd || !m1 || !m2 || !dYsrc || !Gsum || !wgs || B < 1 || Nsrc < 1 ||
        O < 1 || P < 1 || P > 256 || C0 < 1)
        return GRIDGCN_EINVAL;
    if (!workspace || workspace_bytes < gg_edge_lin0_sparse_workspace(B, Nsrc, C0))
        return GRIDGCN_EWORKSPACE;
    const int rc = gg_edge_lin0_bwd_sparse(nebidx, att16, amax, gval, zsel, Ysrc, Wg, b, scale,
                                           shift, mean, rstd, m1, m2, B, Nsrc, O, P, C0, dYsrc,
                                           (float *)Gsum, wgs, nullptr, workspace, (hipStream_t)stream, 1);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_edge_lin0_dwg(const double *wgs, const double *gg, const float *T, const float *wgb,
                          const float *scale, const float *mean, const float *rstd,
                          const float *m1, const float *m2, int C0, float *dW, int ld,
                          void *stream)
{
    if (!wgs || !gg || !T || !wgb || !scale || !mean || !rstd || !m1 || !m2 || !dW || C0 < 1 ||
        ld < 3)
        return GRIDGCN_EINVAL;
    return gg_edge_lin0_dwg(wgs, gg, T, wgb, scale, mean, rstd, m1, m2, C0, dW, ld,
                            (hipStream_t)stream);
}

int gridgcn_att_max_eval(const float *Z1, const float *scale1, const float *shift1, const float *W2,
                         const float *b2, const float *scale_a, const float *shift_a,
                         const float *Ysrc, const int32_t *nebidx, const float *att16,
                         const float *Wg, const float *b, const float *scale_p,
                         const float *shift_p, int B, int Nsrc, int O, int P, int C, float *agg,
                         int ld_agg, void *stream)
{
    if (!Z1 || !scale1 || !shift1 || !W2 || !b2 || !scale_a || !shift_a || !Ysrc || !nebidx ||
        !att16 || !b || !scale_p || !shift_p || !agg || B < 1 || Nsrc < 1 || O < 1 || P < 1 ||
        ld_agg < C)
        return GRIDGCN_EINVAL;
    // the up layers' shape: the tile kernel of the training forward, without its arg max and saved pre-activations
    if (gg_get_att_eval_tile() && gg_att_fwd_ok((long long)B * O, O, P, 32, C, ld_agg, (long long)B * Nsrc))
        return gg_att_pairmax_args(Ysrc, nebidx, att16, Wg, b, B, Nsrc, O, Z1, scale1, shift1, W2, b2, scale_p, shift_p,
                                   scale_a, shift_a, (long long)B * O, agg, ld_agg, nullptr, nullptr,
                                   (hipStream_t)stream);
    GGAttEval p;
    p.Z1 = Z1; p.s1 = scale1; p.h1 = shift1; p.W2 = W2; p.b2 = b2; p.sa = scale_a; p.ha = shift_a;
    p.Ysrc = Ysrc; p.nebidx = nebidx; p.att16 = att16; p.Wg = Wg; p.bp = b; p.sp = scale_p;
    p.hp = shift_p; p.out = agg; p.E = (long long)B * O * P; p.P = P; p.O = O; p.Nsrc = Nsrc;
    p.B = B; p.ldo = ld_agg;
    const int rc = gg_att_max_eval(p, C, (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_ball_knn_grid_workspace_bytes(int B, int m, size_t *bytes)
{
    if (!bytes || B < 1 || m < 1) return GRIDGCN_EINVAL;
    *bytes = gg_ball_grid_workspace(B, m);
    return GRIDGCN_OK;
}

int gridgcn_ball_knn_grid(const float *unknown, const float *known, const int32_t *downnum,
                          const int32_t *upnum, int B, int n, int m, int k, float radius,
                          int32_t *idx, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!unknown || !known || !downnum || !upnum || !idx || B < 1 || n < 1 || m < 1 || k < 1 ||
        k > 6 || !(radius >= 0.f) || (long long)B * n >= (1ll << 31))
        return GRIDGCN_EINVAL;
    if (!workspace || workspace_bytes < gg_ball_grid_workspace(B, m)) return GRIDGCN_EWORKSPACE;
    const int rc = gg_ball_knn_grid(unknown, known, downnum, upnum, B, n, m, k, radius, idx,
                                    workspace, (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_ball_knn_grid_ld(const float *unknown, int ldu, const float *known, int ldk,
                             const int32_t *downnum, const int32_t *upnum, int B, int n, int m, int k,
                             float radius, int zero_tail, int32_t *idx, void *workspace,
                             size_t workspace_bytes, void *stream)
{
    if (!unknown || !known || !downnum || !upnum || !idx || B < 1 || n < 1 || m < 1 || k < 1 ||
        k > 6 || !(radius >= 0.f) || (long long)B * n >= (1ll << 31) || ldu < 3 || ldk < 3)
        return GRIDGCN_EINVAL;
    if (!workspace || workspace_bytes < gg_ball_grid_workspace(B, m)) return GRIDGCN_EWORKSPACE;
    const int rc = gg_ball_knn_grid(unknown, known, downnum, upnum, B, n, m, k, radius, idx,
                                    workspace, (hipStream_t)stream, ldu, ldk, zero_tail ? 1 : 0);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

int gridgcn_cat_mask(const float *a, int lda, int ca, const float *b, int ldb, int cb,
                     const float *mask, float *out, int ldo, float *out2, int ldo2, long long E,
                     void *stream)
{
    if (!out || E < 1 || ca < 0 || cb < 0 || ca + cb < 1 || ldo < ca + cb || (ca && (!a || lda < ca)) ||
        (b && ldb < cb) || (out2 && ldo2 < ca + cb))
        return GRIDGCN_EINVAL;
    return gg_cat_mask(a, lda, ca, b, ldb, cb, mask, out, ldo, out2, ldo2, E, (hipStream_t)stream);
}

int gridgcn_mask_sum(const float *g1, int ld1, const float *g2, int ld2, int col0, int C,
                     const float *mask, float *out, long long E, void *stream)
{
    if (!out || E < 1 || C < 1 || col0 < 0 || (!g1 && !g2) || (g1 && ld1 < col0 + C) ||
        (g2 && ld2 < col0 + C))
        return GRIDGCN_EINVAL;
    return gg_mask_sum(g1, ld1, g2, ld2, col0, C, mask, out, E, (hipStream_t)stream);
}

int gridgcn_adam_step(float *const *params, const float *const *grads, const long long *sizes,
                      const long long *mchunk, int n, float *m, float *v, int32_t *state, float lr,
                      const float *lr_dev, float beta1, float beta2, float eps, float weight_decay,
                      int mode, void *stream)
{
    if (n < 0 || !m || !v || !state || (n && (!params || !grads || !sizes || !mchunk)) ||
        !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) || (mode != 0 && mode != 1))
        return GRIDGCN_EINVAL;
    for (int i = 0; i < n; i++)
        if (!params[i] || !grads[i] || sizes[i] < 0 || sizes[i] >= (1ll << 32) || mchunk[i] < 0 ||
            mchunk[i] >= (1ll << 32) || ((uintptr_t)params[i] & 3) || ((uintptr_t)grads[i] & 3))
            return GRIDGCN_EINVAL;
    return gg_adam_step(params, grads, sizes, mchunk, n, m, v, state, lr, lr_dev, beta1, beta2, eps,
                        weight_decay, mode, (hipStream_t)stream) ? GRIDGCN_ELAUNCH : GRIDGCN_OK;
}

int gridgcn_take_backward_workspace_bytes(int B, int N, int M, size_t *bytes)
{
    if (!bytes || B < 1 || N < 1 || M < 1) return GRIDGCN_EINVAL;
    *bytes = gg_take_bwd_sorted_workspace(B, N, M);
    return GRIDGCN_OK;
}

int gridgcn_edge_inputs_rows_backward(const float *grad_nf, int nf_stride, const int32_t *nebidx,
                                      int B, int Nsrc, int Cs, int O, int P, float *grad_src,
                                      void *workspace, size_t workspace_bytes, void *stream)
{
    if (!grad_nf || !nebidx || !grad_src || B < 1 || Nsrc < 1 || O < 1 || P < 1 || Cs <= 4 ||
        nf_stride < Cs - 4)
        return GRIDGCN_EINVAL;
    // features are the first Cs-4 columns of a row; xyz/w receive no gradient
    if (workspace) {
        if (workspace_bytes < gg_take_bwd_sorted_workspace(B, Nsrc, O * P)) return GRIDGCN_EWORKSPACE;
        const int rc = gg_take_bwd_sorted(grad_nf, nebidx, B, Nsrc, Cs - 4, O * P, grad_src + 4,
                                          nf_stride, Cs, workspace, (hipStream_t)stream);
        if (rc != 1) return rc;          // 1: shape not covered by the sorted path
    }
    return gg_batch_take_backward(grad_nf, nebidx, B, Nsrc, Cs - 4, O * P, grad_src + 4, nf_stride,
                                  Cs, (hipStream_t)stream);
}

int gridgcn_gridconv_forward(const float *src, const int32_t *nebidx, const float *cent,
                             int cent_stride, int B, int Nsrc, int Cs, int O, int P,
                             int has_feats, int localfdim, int npt, const gridgcn_conv_layer *pt,
                             const gridgcn_conv_layer *att, float *out, void *stream)
{
    if (!src || !nebidx || !cent || !pt || !att || !out) return GRIDGCN_EINVAL;
    if (B < 1 || Nsrc < 1 || O < 1 || P < 1 || P > 128 || Cs < 4 || npt < 1 || npt > 4)
        return GRIDGCN_EINVAL;
    if (has_feats && Cs == 4) return GRIDGCN_EINVAL;
    GGConvParams p;
    p.src = src; p.nebidx = nebidx; p.cent = cent; p.out = out; p.cent_stride = cent_stride;
    p.B = B; p.Nsrc = Nsrc; p.Cs = Cs; p.O = O; p.P = P;
    p.has_feats = has_feats; p.localfdim = localfdim; p.npt = npt;
    // LDS row of the first pt layer = the gathered source row with columns 0..3 replaced by
    // (geo_vec, 0): K0 = Cs rounded up to a multiple of 4 (the packer zero-fills unused rows)
    auto r4 = [](int x) { return (x + 3) & ~3; };
    int amax = r4(Cs);
    auto conv = [](const gridgcn_conv_layer &s, GGConvLayer *d) -> bool {
        if (!s.W || !s.b || s.K < 4 || (s.K & 3)) return false;
        if (s.ldw != 32 && s.ldw != 64 && s.ldw != 128 && s.ldw != 256) return false;
        if (s.cout < 1 || s.cout > s.ldw) return false;
        d->W = s.W; d->b = s.b; d->K = s.K; d->ldw = s.ldw; d->cout_real = s.cout; d->pad_ = 0;
        return true;
    };
    int kin = amax;
    for (int l = 0; l < npt; l++) {
        if (!conv(pt[l], &p.pt[l]) || p.pt[l].K != kin) return GRIDGCN_EINVAL;
        if (l < npt - 1) {
            if (p.pt[l].ldw > 128) return GRIDGCN_EINVAL;   // in-place layers: <= 128 wide
            if (p.pt[l].ldw > amax) amax = p.pt[l].ldw;
            kin = r4(p.pt[l].cout_real);
        }
    }
    for (int l = npt; l < 4; l++) p.pt[l] = p.pt[0];
    if (!conv(att[0], &p.att[0]) || !conv(att[1], &p.att[1])) return GRIDGCN_EINVAL;
    if (p.att[0].K != 12 || p.att[0].ldw > 128 || p.att[1].K != r4(p.att[0].cout_real))
        return GRIDGCN_EINVAL;
    if (p.att[1].ldw != p.pt[npt - 1].ldw || p.att[1].cout_real != p.pt[npt - 1].cout_real)
        return GRIDGCN_EINVAL;
    int tmax = p.att[0].ldw > 12 ? p.att[0].ldw : 12;
    p.lda = amax | 1;          // odd row strides: conflict-free ds_read_b32 down a column
    p.ldt = tmax | 1;
    int rc = gg_gridconv_forward(p, (hipStream_t)stream);
    return rc == 1 ? GRIDGCN_EINVAL : rc;
}

}  // extern "C"
