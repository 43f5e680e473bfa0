/*
 * gridgcn.h -- C ABI of libgridgcn_hip.so: the MI355X (gfx950) implementation of the
 * Grid-GCN index operators (Coverage-Aware Grid Query family) and of the GridConv
 * neighbour gather.
 *
 * The reference has no C ABI: its operators are registered into MXNet's C++ operator
 * registry by static initialisers of gridifyop/additional.so
 *   MXNET_REGISTER_OP_PROPERTY(Gridify, GridifyProp)        gridify.cc:63-67
 *   MXNET_REGISTER_OP_PROPERTY(GridifyUp, GridifyUpProp)    gridify_up.cc:62-68
 *   MXNET_REGISTER_OP_PROPERTY(GridifyKNN, GridifyKNNProp)  gridifyknn.cc:63-67
 *   NNVM_REGISTER_OP(_contrib_BallKNN)                      ball_k_nn.cc:14
 *   NNVM_REGISTER_OP(_contrib_KNN)                          k_nn.cc:14
 * and each ends in a host function taking raw device pointers:
 *   GridifyForward<gpu>     gridify.cu:294-413
 *   GridifyUpForward<gpu>   gridify_up.cu:228-323
 *   GridifyKNNForward<gpu>  gridifyknn.cu:336-455
 *   BallKNNForward<gpu>     ball_k_nn-inl.h:96-116
 *   KNNForward<gpu>         k_nn-inl.h:94-113
 * The entry points below are drop-ins for exactly those host functions: same tensors, same
 * layouts (row-major, contiguous), same dtypes, same attribute meaning.  INTEGRATION.md
 * shows the MXNet-side Forward() stub a maintainer would write on top of them.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HBM), contiguous, 16-byte aligned for float4 rows;
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it:
 *     no allocation, no host synchronisation, no default-stream use (capturable in a hipGraph);
 *   - scratch comes from the caller: query the size with *_workspace_bytes, pass any buffer of
 *     at least that size (256-byte aligned); it can be reused by later calls on the same stream;
 *   - return value: 0 = ok, otherwise a GRIDGCN_E* code; gridgcn_strerror() explains;
 *     no C++ exception ever crosses this boundary;
 *   - outputs are fully written by the call (the reference pre-fills them in
 *     GridifyOp::Forward, gridify-inl.h:117-121; that fill is folded into the kernels).
 */
#ifndef GRIDGCN_H_
#define GRIDGCN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRIDGCN_OK 0
#define GRIDGCN_EINVAL 1      /* bad argument (shape/attribute out of the supported domain) */
#define GRIDGCN_EWORKSPACE 2  /* workspace too small / null */
#define GRIDGCN_ELAUNCH 3     /* hipGetLastError() != hipSuccess after a launch */

/* Attributes of Gridify / GridifyKNN / GridifyUp = GridifyParam (gridify-inl.h:58-87) and
 * GridifyUpParam (gridify_up-inl.h:58-81).  `seed` replaces the reference's
 * gettimeofday().tv_usec (gridify.cu:377-379); seed 0 == the fast_apprxmt fixed-seed build. */
typedef struct gridgcn_grid_params {
    int32_t max_p_grid;     /* P: neighbours kept per centre / per voxel bucket (<= 128)        */
    int32_t max_o_grid;     /* O: centres per cloud (GridifyUp: number of up points M)          */
    int32_t kernel_size;    /* k: odd, <= 7; neighbourhood is k^3 voxels                         */
    int32_t stride;         /* accepted and ignored, as in the reference kernels                */
    int32_t loc;            /* 1: centre xyz = weighted mean of the centre voxel's points       */
    float coord_shift[3];
    float voxel_size[3];
    int32_t grid_size[3];   /* gx*gy*gz < 2^24 and B*gx*gy*gz < 2^31                             */
    uint64_t seed;
    const uint64_t *seed_dev; /* optional DEVICE scalar added to `seed` when the kernels run (NULL:
                               * none).  Lets a captured hipGraph of the operator draw a fresh
                               * sample at every replay: the caller bumps the scalar on the stream. */
} gridgcn_grid_params;

const char *gridgcn_strerror(int code);
/* library/ABI version, bumped on any signature change */
int gridgcn_abi_version(void);  /* 2: seed_dev / drop_seed_dev; 3: one-byte arg-max tensors;
                                 * 4: gridgcn_set_option, Z-less attention pair removed, the library
                                 *    reads nothing from the process environment
                                 * 5: gridgcn_pairmax_bwd_masked, gridgcn_att_bwd_noz, gridgcn_gemm_bias, options 3 / 4
                                 * 6: gridgcn_pack_desc.wgb / geo, gridgcn_adam_step, gridgcn_cat_mask,
                                 *    gridgcn_mask_sum, gridgcn_ball_knn[_grid]_ld, gridgcn_bn_finalize_tail,
                                 *    gridgcn_softmax_ce_loss, gridgcn_colsum_f32, gridgcn_edge_geo_forward, gridgcn_edge_lin0_backward_sparse_geo, gridgcn_linear_fwd_direct_fin,
                                 *    gridgcn_linear_fwd_direct_drop, gridgcn_linear_dw_drop, GRIDGCN_OPT_PAIRMAX_SPLIT; psums of a dX launch with
                                 *    nbn > 0 is [2][nbn]
                                 * 7: GRIDGCN_OPT_ATT_NZ_V2, GRIDGCN_OPT_BWD_FUSED128 (gridgcn_linear_bwd may take the one-pass
                                 *    kernel: dX / dW / sums in other summation orders); gridgcn_gemm_small_workspace_bytes is
                                 *    bounded (~16 MB) whatever the row count
                                 * 8: gridgcn_att_bn2_moments, gridgcn_att_pairmax_fwd (+ _workspace_bytes), GRIDGCN_OPT_ATT_EVAL_TILE
                                 * 9: gridgcn_att_pairmax_fwd_supported, gridgcn_att_bwd_noz_mom (+ _supported), gridgcn_att_moments_offset */

/* Kernel-selection options (process-wide, read at launch time; for A/B tests -- the defaults are
 * what is measured and shipped).  set: 0 ok / GRIDGCN_EINVAL for an unknown option; get: -1. */
#define GRIDGCN_OPT_ATT_BWD_FUSED 0  /* [1] one-pass backward of the attention conv (dX, previous
                                      *     layer's BN sums and dW from one read of Z); 0 = the
                                      *     separate dX and dW kernels */
#define GRIDGCN_OPT_INDEX_SLAB_SHIFT 1 /* [0] voxel-index build: added to log2(slabs per cloud), -4..4  */
#define GRIDGCN_OPT_INDEX_CHUNK 2      /* [0 = automatic] voxel-index build: points per chunk, 1024 / 2048 /
                                        *     4096.  Both only move work between the build's kernels;
                                        *     results are identical for every setting. */
#define GRIDGCN_OPT_INDEX_SMALL 3      /* [1] clouds of <= 4096 points (and max_o_grid <= 4096): the whole
                                        *     voxel-index build as ONE launch, one workgroup per cloud, all
                                        *     in LDS; 0 = the three-launch build for every size.  Identical
                                        *     results.  The workspace size depends on it: set it before
                                        *     asking for *_workspace_bytes, not between that and the call. */
#define GRIDGCN_OPT_COL_SPLIT 4        /* [1] forward / input-gradient GEMM kernels of layers with <= 16 K rows:
                                        *     the output column tiles go to separate workgroups (a 32-row
                                        *     tile is one serial MFMA chain per wave; such layers have too few
                                        *     tiles to fill the chip); 0 = one wave owns whole rows.  Same
                                        *     arithmetic per element: identical results. */
#define GRIDGCN_OPT_PAIRMAX_SPLIT 5    /* [0] lanes per (centre, 4 channels) of gridgcn_pairmax_fwd: 0 = chosen from
                                        *     the layer's size, 1 / 2 / 4 / 8 forced (tuning; the first arg max is
                                        *     exact for every setting). */
#define GRIDGCN_OPT_ATT_NZ_V2 6        /* [1] gridgcn_att_bwd_noz: the round-5 tile loop (no per-tile divisions,
                                        *     range-checked buffer streams); 0 = the round-4 kernel.  Identical
                                        *     results, word for word (tests/test_gpu_train_ops.py). */
#define GRIDGCN_OPT_BWD_FUSED128 7     /* [1] gridgcn_linear_bwd of a 128-output layer with 128 / 256 inputs, dense
                                        *     gradient, E % 128 == 0, E >= 32768: dX, dW and the sums of the layer
                                        *     in front from ONE pass over Z and dY (csrc/gridgcn_bwdfused.hip);
                                        *     0 = the separate dX and dW kernels; 2 = only layers of 128 inputs (the
                                        *     256-input update conv takes two launches of it: worth 0.03 ms of a cfg4
                                        *     step, the 128-input fc1 0.1 ms).  Same terms, other summation orders. */
#define GRIDGCN_OPT_ATT_EVAL_TILE 8     /* [1] gridgcn_att_max_eval at the up layers' shape (P = 5, 32 -> 128 and the limits of
                                        *     gridgcn_att_pairmax_fwd): the tile kernel of the training forward without
                                        *     its arg max and saved pre-activations (0.41 ms against 1.1 at cfg4 up2);
                                        *     0 = the general kernel for every shape.  Same terms, last-bit differences. */
int gridgcn_set_option(int option, int value);
int gridgcn_get_option(int option);

/* Precision of the contraction inside the training GEMM kernels (gridgcn_linear_fwd_direct,
 * gridgcn_linear_dx, the direct dW kernel behind gridgcn_linear_bwd): 0 (default) = exact fp32
 * (v_mfma_f32_32x32x2_f32), the parity path; 1 = bf16 operands (v_mfma_f32_32x32x16_bf16, round to
 * nearest even at the register level) with fp32 accumulation -- tensors in memory, BatchNorm
 * statistics and every epilogue stay fp32.  BASELINE configs[2] "bf16 MLP / fp32 indices".
 * PROCESS-WIDE and read at launch time: set it before enqueuing the kernels it should affect. */
int gridgcn_set_mlp_precision(int bf16);
int gridgcn_get_mlp_precision(void);

/* ---- Gridify : replaces GridifyForward<gpu>, gridify.cu:294-413 -------------------------------
 * in : data[B,N,4] f32 (x,y,z,w)   actual_numpoints[B] i32
 * out: nebidx[B,O,P] i32  nebidxmsk[B,O,P] f32  cent[B,O,4] f32  centmsk[B,O] f32
 *      actual_centnum[B] i32                                   (shapes: gridify-inl.h:190-195) */
int gridgcn_gridify_workspace_bytes(int B, int N, const gridgcn_grid_params *p, size_t *bytes);
int gridgcn_gridify(const float *data, const int32_t *actual_numpoints, int B, int N,
                    const gridgcn_grid_params *p,
                    int32_t *nebidx, float *nebidxmsk, float *cent, float *centmsk,
                    int32_t *actual_centnum,
                    void *workspace, size_t workspace_bytes, void *stream);

/* ---- Gridify_occaware: Gridify with Coverage-Aware Sampling (CAS) of the centre voxels ---------
 * PARITY UNPINNED: the reference holds this operator only as a binary (gridifyop/additional.so,
 * symbols GridifyOp_occaware*, gridify_occaware_sampling; no source, no caller).  Implemented from
 * the paper (Grid-GCN, CVPR 2020, section 3.2 eq. 3-4) under the schedule written down in
 * oracle/gridgcn_oracle.c: the RVS sample of gridgcn_gridify gives the incumbents; every other
 * occupied voxel, in order of first appearance, challenges one random incumbent and replaces it
 * when H_add > H_rmv (beta >= 0 weighs the over-coverage penalty; beta = 0: pure coverage).
 * Same inputs / outputs / attributes as gridgcn_gridify; max_o_grid <= 16384, grid_size[j] <= 1023. */
int gridgcn_gridify_occaware_workspace_bytes(int B, int N, const gridgcn_grid_params *p,
                                             size_t *bytes);
int gridgcn_gridify_occaware(const float *data, const int32_t *actual_numpoints, int B, int N,
                             const gridgcn_grid_params *p, float beta,
                             int32_t *nebidx, float *nebidxmsk, float *cent, float *centmsk,
                             int32_t *actual_centnum,
                             void *workspace, size_t workspace_bytes, void *stream);

/* ---- Gridify, `fast_rand` build variant (gridifyop/fast_rand/gridify.cu:126-272) ---------------
 * Same operator, different sampling: every point enters the bucket of all k^3 voxels around its
 * own (reservoir past P seeded with the reference's thread index, so `seed` is not used), the
 * centres are the first max_o_grid occupied voxels in order of first appearance, a centre reads the
 * bucket of its own voxel only; loc == 0: centre = weighted mean of the picked points.  Results
 * are those of the reference's kernels under the canonical schedule S0; B*N*k^3 < 2^31. */
int gridgcn_gridify_fast_rand_workspace_bytes(int B, int N, const gridgcn_grid_params *p,
                                              size_t *bytes);
int gridgcn_gridify_fast_rand(const float *data, const int32_t *actual_numpoints, int B, int N,
                              const gridgcn_grid_params *p,
                              int32_t *nebidx, float *nebidxmsk, float *cent, float *centmsk,
                              int32_t *actual_centnum,
                              void *workspace, size_t workspace_bytes, void *stream);

/* Measurement helper: `iters` back-to-back gridgcn_gridify calls on `stream`, bracketed by HIP
 * events recorded on that stream; *ms_per_call = average device time of one call (all of its
 * launches, no host work in between except the launches themselves).  Synchronises the stream. */
int gridgcn_gridify_timed(const float *data, const int32_t *actual_numpoints, int B, int N,
                          const gridgcn_grid_params *p,
                          int32_t *nebidx, float *nebidxmsk, float *cent, float *centmsk,
                          int32_t *actual_centnum,
                          void *workspace, size_t workspace_bytes, void *stream, int iters,
                          float *ms_per_call);

/* ---- GridifyKNN : replaces GridifyKNNForward<gpu>, gridifyknn.cu:336-455 ----------------------
 * same tensors as Gridify; neighbours = exact top-P by distance to the voxel centre over
 * Chebyshev shells (gridifyknn.cu:231-332). */
int gridgcn_gridify_knn_workspace_bytes(int B, int N, const gridgcn_grid_params *p, size_t *bytes);
int gridgcn_gridify_knn(const float *data, const int32_t *actual_numpoints, int B, int N,
                        const gridgcn_grid_params *p,
                        int32_t *nebidx, float *nebidxmsk, float *cent, float *centmsk,
                        int32_t *actual_centnum,
                        void *workspace, size_t workspace_bytes, void *stream);

/* ---- GridifyUp : replaces GridifyUpForward<gpu>, gridify_up.cu:228-323 ------------------------
 * in : downdata[B,Nd,4] f32  updata[B,O,4] f32  down_actual_numpoints[B] i32
 *      up_actual_numpoints[B] i32            (O = max_o_grid: updata is indexed with stride O,
 *                                             gridify_up.cu:196)
 * out: nebidx[B,O,P] i32  nebidxmsk[B,O,P] f32                (gridify_up-inl.h:183-184) */
int gridgcn_gridify_up_workspace_bytes(int B, int Nd, const gridgcn_grid_params *p, size_t *bytes);
int gridgcn_gridify_up(const float *downdata, const float *updata,
                       const int32_t *down_actual_numpoints, const int32_t *up_actual_numpoints,
                       int B, int Nd, const gridgcn_grid_params *p,
                       int32_t *nebidx, float *nebidxmsk,
                       void *workspace, size_t workspace_bytes, void *stream);

/* ---- BallKNN / KNN : replace BallKNNForward<gpu> (ball_k_nn-inl.h:96-116) and
 *      KNNForward<gpu> (k_nn-inl.h:94-113) ------------------------------------------------------
 * in : unknown[B,n,3] f32  known[B,m,3] f32  downnum[B] i32  upnum[B] i32
 * out: idx[B,n,k] i32; -1 = no neighbour; rows >= upnum[b] are NOT written (as the reference).
 * BallKNN: k <= 6 (best[6], ball_k_nn-inl.h:63).  KNN: k <= 64 here. */
int gridgcn_ball_knn(const float *unknown, const float *known, const int32_t *downnum,
                     const int32_t *upnum, int B, int n, int m, int k, float radius,
                     int32_t *idx, void *stream);
/* gridgcn_ball_knn_grid: the same result as gridgcn_ball_knn (bit for bit: the reference's strict-<
 * insertion over ascending indices keeps the k smallest by (distance, index), whatever the
 * traversal), through a uniform cell grid (cell >= 1.001 * radius, <= 24^3 cells) built over the
 * cloud's known points: 27 cells instead of m points per query.  k <= 6.
 * workspace: gridgcn_ball_knn_grid_workspace_bytes(B, m), 16-byte aligned (it holds float4 records;
 * a misaligned base is refused with GRIDGCN_EINVAL). */
int gridgcn_ball_knn_grid_workspace_bytes(int B, int m, size_t *bytes);
int gridgcn_ball_knn_grid(const float *unknown, const float *known, const int32_t *downnum,
                          const int32_t *upnum, int B, int n, int m, int k, float radius,
                          int32_t *idx, void *workspace, size_t workspace_bytes, void *stream);
/* gridgcn_ball_knn_ld / gridgcn_ball_knn_grid_ld: the same with the coordinates read in place from wider rows -- unknown[B,n]
 * rows of ldu floats, known[B,m] rows of ldk floats (x, y, z first; the [B,n,4+C] point rows of the
 * up path, ggcn_models_g.py:204-205, without the two packing copies) -- and, zero_tail != 0, rows
 * >= upnum[b] of idx written as 0 (for a caller that hands over uninitialised memory). */
int gridgcn_ball_knn_ld(const float *unknown, int ldu, const float *known, int ldk,
                        const int32_t *downnum, const int32_t *upnum, int B, int n, int m, int k,
                        float radius, int zero_tail, int32_t *idx, void *stream);
int gridgcn_ball_knn_grid_ld(const float *unknown, int ldu, const float *known, int ldk,
                             const int32_t *downnum, const int32_t *upnum, int B, int n, int m, int k,
                             float radius, int zero_tail, int32_t *idx, void *workspace,
                             size_t workspace_bytes, void *stream);
int gridgcn_knn(const float *unknown, const float *known, const int32_t *downnum,
                const int32_t *upnum, int B, int n, int m, int k,
                int32_t *idx, void *stream);

/* ---- batch_take : replaces the MXNet graph of utils/ops.py:78-93 (batch_take_g) ---------------
 * out[b, j, :] = data[clip(index[b, j] + b*N, 0, B*N-1), :]    (mx.sym.take default mode='clip')
 * data[B,N,C] f32, index[B,M] i32, out[B,M,C] f32.  Backward = scatter-add of grad_out into
 * grad_data (must be zero-filled by the caller). */
int gridgcn_batch_take(const float *data, const int32_t *index, int B, int N, int C, int M,
                       float *out, void *stream);
int gridgcn_batch_take_backward(const float *grad_out, const int32_t *index, int B, int N, int C,
                                int M, float *grad_data, void *stream);
/* the same sums as a sorted segmented sum (counting sort of the indices by destination row, whole
 * gradient rows summed per run; atomics only where a 256-edge chunk cuts a run) -- 3-5x faster than
 * the scatter-add when many edges share a destination.  workspace:
 * gridgcn_take_backward_workspace_bytes(B, N, M); falls back to the scatter-add for row widths the
 * sorted kernel does not take.  grad_data zero-filled by the caller.  PRECONDITION: every index
 * lies in [-1, N-1] (the range the index operators produce; -1 clips into the previous cloud's last
 * row as in the reference) -- the counting sort bins by destination row within the cloud. */
int gridgcn_batch_take_backward_sorted(const float *grad_out, const int32_t *index, int B, int N,
                                       int C, int M, float *grad_data, void *workspace,
                                       size_t workspace_bytes, void *stream);

/* ---- edge inputs of sub_g_update (training path) ----------------------------------------------
 * One pass instead of batch_take_g + slice_axis + tile + sub + sqrt(sum(square)) + concat
 * (segmentation/models/gcn_module_g_att.py:190-194, 217-218, 242-250):
 *   nf [B,O,P,cin]  = geo_vec (no features) | features (localfdim == 0) | concat(geo_vec, features)
 *   att[B,O,P,10]   = (geo_dist, geo_vec, centre xyz, neighbour xyz)           (attfdim == 10)
 * Backward: only the feature columns of src receive gradient (scatter-add, grad_src zero-filled
 * by the caller); positions come from the non-differentiable index ops (gridify-inl.h:227-231). */
int gridgcn_edge_inputs(const float *src, const int32_t *nebidx, const float *cent,
                        int cent_stride, int B, int Nsrc, int Cs, int O, int P, int has_feats,
                        int localfdim, float *nf, float *att, void *stream);
int gridgcn_edge_inputs_backward(const float *grad_nf, const int32_t *nebidx, int B, int Nsrc,
                                 int Cs, int O, int P, int has_feats, int localfdim,
                                 float *grad_src, void *stream);

/* "rows" variant for the MFMA training kernels: the same values, laid out so that every row is read
 * and written with 16-byte accesses --
 *   nf [B,O,P,nf_stride] = features (Cs-4 columns, a multiple of 4) | geo_vec (if the layer has one)
 *                          | zeros;  nf_stride a multiple of 8
 *   att16[B,O,P,16]      = the 10 att_vec channels | 6 zeros
 * (the first conv's weight columns are permuted/padded to match by gridgcn_pack_linear's `rot`).
 * Backward: grad_src[.., 4:] += the first Cs-4 columns of grad_nf rows. */
int gridgcn_edge_inputs_rows(const float *src, const int32_t *nebidx, const float *cent,
                             int cent_stride, int B, int Nsrc, int Cs, int O, int P, int has_feats,
                             int localfdim, int nf_stride, float *nf, float *att16, void *stream);
int gridgcn_edge_inputs_rows_backward(const float *grad_nf, int nf_stride, const int32_t *nebidx,
                                      int B, int Nsrc, int Cs, int O, int P, float *grad_src,
                                      void *workspace, size_t workspace_bytes, void *stream);
/* workspace (gridgcn_take_backward_workspace_bytes(B, Nsrc, O*P)) != NULL selects the SORTED
 * backward: the edges of each cloud are ordered by destination row with a counting sort of nebidx
 * and summed run by run from whole gradient rows -- no scatter atomics except at the <= 2 run
 * fragments a 256-edge chunk can cut.  NULL: LDS-privatised scatter-add.  grad_src zero-filled by
 * the caller in both cases.  Sums are reproducible to fp32 round-off, not bit for bit. */
int gridgcn_take_backward_workspace_bytes(int B, int N, int M, size_t *bytes);

/* ---- first conv of the point MLP WITHOUT the gathered tensor ---------------------------------
 * The first 1x1 conv of sub_g_update's point MLP acts on concat(geo_vec, gathered features)
 * (gcn_module_g_att.py:190-194, 242-250, 135).  Linear + gather commute:
 *   Z0[e] = Ysrc[src(e)] + Wg * geo_vec(e) + b,   Ysrc[B*Nsrc, C0] = features * Wf^T (once per point)
 * forward : Z0[B*O*P, C0], att16[B*O*P, 16] and sums[2*C0] += (sum z, sum z^2) (BatchNorm statistics;
 *           zeroed by the caller).  Ysrc NULL = no feature term, Wg[3][C0] NULL = no geo term.
 *           Z0 NULL = not stored; Z0 and sums both NULL (evaluation) = only att16 is produced.
 * backward: dZ0 = BatchNorm/ReLU backward of the upstream gradient (dense dY[E,C0], or NULL and the
 *           sparse (amax, gval)[B*O, C0] of gridgcn_pairmax_bwd) formed on the fly, summed per source
 *           row over the edges sorted by destination -> dYsrc[B*Nsrc, C0] (zeroed by the caller),
 *           dWg[3][C0] += sum_e geo_vec(e) dZ0[e] (fp64, zeroed by the caller; NULL = skip).
 *           workspace: gridgcn_take_backward_workspace_bytes(B, Nsrc, O*P). */
int gridgcn_edge_lin0_forward(const float *Ysrc, const float *src, const int32_t *nebidx,
                              const float *cent, int cent_stride, int B, int Nsrc, int Cs, int O,
                              int P, int C0, const float *Wg, const float *b, float *Z0,
                              float *att16, double *sums, void *stream);
/* gridgcn_edge_geo_forward: att16 and the BatchNorm statistics of a single-layer point MLP (Z0 never
 *   stored) WITHOUT the edge x channel pass: z0[e,c] = (Ysrc[n(e),c] + b[c]) + Wg[:,c] . geo(e) is affine
 *   in per-source and per-edge quantities, so sum z0 and sum z0^2 follow from cnt(n) = #edges of source
 *   n, G(n) = sum of their geo_vec and GG = sum_e geo geo^T (csrc/gridgcn_edgelin.hip) -- 8 K source
 *   rows x C0 instead of 3.3 M edges x C0 at BASELINE configs[3]'s last up layer.
 *   out: att16[E][16] as gridgcn_edge_lin0_forward; Gsum[B*Nsrc][4] = (G, cnt) per source row;
 *   gg[12] += (GG[9], sum geo[3]) and sums[2*C0] += (sum z0, sum z0^2), both fp64, zeroed by the caller.
 *   (Nsrc + 1) * 28 bytes of LDS per cloud: GRIDGCN_EINVAL beyond 150 KB (use gridgcn_edge_lin0_forward). */
int gridgcn_edge_geo_forward_workspace_bytes(int B, int Nsrc, int O, int P, size_t *bytes);
int gridgcn_edge_geo_forward(const float *Ysrc, const float *src, const int32_t *nebidx, const float *cent,
                             int cent_stride, int B, int Nsrc, int Cs, int O, int P, int C0,
                             const float *Wg, const float *b, float *att16, float *Gsum, double *gg,
                             double *sums, void *workspace, size_t workspace_bytes, void *stream);
/* Z0 may be NULL in both calls: the forward then only produces the statistics and att16, and the
 * consumers (gridgcn_pairmax_fwd_src, the backward) recompute Z0 from (Ysrc, Wg, b) with the same
 * operation order, i.e. bit-identical -- the [E, C0] tensor never exists (single-layer point MLPs). */
int gridgcn_pairmax_fwd_src(const float *Ysrc, const int32_t *nebidx, const float *att16,
                            const float *Wg, const float *b, int B, int Nsrc, int O,
                            const float *Za, const float *scale_p, const float *shift_p,
                            const float *scale_a, const float *shift_a, long long ncent, int P,
                            int C, float *agg, int ld_agg, uint8_t *amax, float *zsel,
                            void *stream);
int gridgcn_pairmax_fwd_src_z(const float *Ysrc, const int32_t *nebidx, const float *att16,
                              const float *Wg, const float *b, int B, int Nsrc, int O,
                              const void *Za, int za_bf16, const float *scale_p, const float *shift_p,
                              const float *scale_a, const float *shift_a, long long ncent, int P,
                              int C, float *agg, int ld_agg, uint8_t *amax, float *zsel,
                              void *stream);
/* Evaluation-mode tail of the edge block in one kernel (csrc/gridgcn_atteval.hip): with every
 * BatchNorm a fixed affine map (scale = gamma*rsqrt(running_var+eps), shift = beta - mean*scale),
 *   agg[o,c] = max_p relu((Ysrc[src(e)][c] + Wg[:,c].geo(e) + b[c]) * scale_p[c] + shift_p[c])
 *                  * relu((W2[c,:] . relu(Z1[e,:]*scale1 + shift1) + b2[c]) * scale_a[c] + shift_a[c])
 * Z1[E,32] = raw output of the first attention conv, W2[C][32] (torch layout) / b2 the second one;
 * C = 64 or 128, P <= 8, agg[B*O][ld_agg].  The [E, C] attention tensor is never materialised. */
int gridgcn_att_max_eval(const float *Z1, const float *scale1, const float *shift1, const float *W2,
                         const float *b2, const float *scale_a, const float *shift_a,
                         const float *Ysrc, const int32_t *nebidx, const float *att16,
                         const float *Wg, const float *b, const float *scale_p,
                         const float *shift_p, int B, int Nsrc, int O, int P, int C, float *agg,
                         int ld_agg, void *stream);
int gridgcn_edge_lin0_backward(const float *Z0, const float *Ysrc, const float *Wg, const float *b,
                               const float *dY, const uint8_t *amax,
                               const float *gval, const float *scale, const float *shift,
                               const float *mean, const float *rstd, const float *m1,
                               const float *m2, const float *att16, const int32_t *nebidx, int B,
                               int Nsrc, int O, int P, int C0, float *dYsrc, double *dWg,
                               void *workspace, size_t workspace_bytes, void *stream);

/* Sparse form of that backward for a SINGLE-layer point MLP (upstream = the max pool: (amax, gval)
 * of gridgcn_pairmax_bwd, zsel = Z0 at the arg max as kept by gridgcn_pairmax_fwd*).  dZ0's dense
 * BatchNorm terms are affine in z0 = Ysrc[src] + Wg geo + b, so their per-source sum needs only the
 * per-source edge count and sum of geo_vec; the ncent*C0 arg-max entries are scattered with LDS
 * atomics.  Outputs: dYsrc[B*Nsrc, C0]; Gsum[B*Nsrc, 4] = (sum geo_vec, count) per source;
 * wgs[3*C0] += sum_(o,c) geo_vec(e*) s[o,c]; gg[12] += (sum geo geo^T [9], sum geo [3]) (fp64, zeroed by
 * the caller) -- from which dWg = wgs + bz (Gsum[:, :3]^T (Ysrc + b) + GG Wg) + (cz - mean bz) sum geo,
 * bz = -scale*rstd*m2, cz = -scale*m1. */
int gridgcn_edge_lin0_backward_sparse_workspace_bytes(int B, int Nsrc, int C0, size_t *bytes);
int gridgcn_edge_lin0_backward_sparse(const int32_t *nebidx, const float *att16,
                                      const uint8_t *amax, const float *gval, const float *zsel,
                                      const float *Ysrc, const float *Wg, const float *b,
                                      const float *scale, const float *shift, const float *mean,
                                      const float *rstd, const float *m1, const float *m2, int B,
                                      int Nsrc, int O, int P, int C0, float *dYsrc, float *Gsum,
                                      double *wgs, double *gg, void *workspace,
                                      size_t workspace_bytes, void *stream);
/* ..._geo: Gsum (and gg) are INPUTS, left by gridgcn_edge_geo_forward of the same layer: the kernel's geo
 * pass over the edges is not run. */
int gridgcn_edge_lin0_backward_sparse_geo(const int32_t *nebidx, const float *att16,
                                          const uint8_t *amax, const float *gval, const float *zsel,
                                          const float *Ysrc, const float *Wg, const float *b,
                                          const float *scale, const float *shift, const float *mean,
                                          const float *rstd, const float *m1, const float *m2, int B,
                                          int Nsrc, int O, int P, int C0, float *dYsrc,
                                          const float *Gsum, double *wgs, void *workspace,
                                          size_t workspace_bytes, void *stream);
/* gridgcn_edge_lin0_dwg: that last formula in one launch.  T[C0][4] = Ysrc^T Gsum (columns 0..2),
 * wgb[4][C0] = (Wg rows, b).  dWg is written transposed into dW[c*ld + j], j = 0..2: the geo_vec
 * columns of the layer's weight gradient [C0][3 + Cf] (ld = 3 + Cf). */
int gridgcn_edge_lin0_dwg(const double *wgs, const double *gg, const float *T, const float *wgb,
                          const float *scale, const float *mean, const float *rstd,
                          const float *m1, const float *m2, int C0, float *dW, int ld,
                          void *stream);

/* ---- training-mode 1x1 conv + BatchNorm + ReLU (utils/ops.py:149-158 conv2d, :141-147 conv1d) ---
 * gridgcn_linear_fwd: Z[E,cout] = act(X[E,cin]) * W + b on fp32 MFMA; act = identity (scale ==
 *   NULL) or the previous layer's BatchNorm+ReLU x -> relu(x*scale[c] + shift[c]) applied while
 *   staging; sums[0:cout] += sum_e Z, sums[cout:2cout] += sum_e Z^2 (fp64, zeroed by the caller):
 *   the batch statistics of THIS layer's BatchNorm.  W packed as for gridgcn_gridconv_forward.
 * gridgcn_bn_relu_apply: Y = relu(Z*scale + shift).
 * gridgcn_bn_relu_bwd_reduce: sums[0:C] += sum dyr, sums[C:2C] += sum dyr*zhat with
 *   dyr = dY * (Z*scale+shift > 0), zhat = (Z-mean)*rstd.  C must divide 256 or be a multiple.
 * gridgcn_bn_relu_bwd_elemt: dZ = scale * (dyr - m1 - zhat*m2)   (m1 = s1/E, m2 = s2/E). */
int gridgcn_linear_fwd(const float *X, long long E, int cin, const float *W, const float *b, int K,
                       int ldw, int cout, const float *scale, const float *shift, float *Z,
                       double *sums, void *stream);
/* *_ld variants: Z with a row stride ldz >= cout (0 = cout): the layer's raw output is written into /
 * read from the left columns of a wider row-major buffer -- update_func's concat(centre features,
 * aggregate), whose consumer applies the BatchNorm+ReLU while loading, so that neither the activated
 * copy nor a separate BatchNorm-backward reduce pass exists (gridgcn_linear_bwd_ld: the
 * register-direct dX / dW kernels only; other shapes return GRIDGCN_EINVAL;
 * nbn: only the first nbn input columns (a multiple of 32; 0 = all) carry a previous BatchNorm -- the
 * dX epilogue reads Aprev and accumulates psums for those alone, and psums is then [2][nbn]: the
 * producer's own table). */
int gridgcn_linear_fwd_ld(const float *X, long long E, int cin, const float *W, const float *b, int K,
                          int ldw, int cout, const float *scale, const float *shift, float *Z,
                          double *sums, int ldz, void *stream);
/* gridgcn_linear_bwd: backward of one (linear -> BatchNorm(batch stats) -> ReLU) layer in ONE pass
 *   over the edges: dZ = scale*(dyr - m1 - zhat*m2) is formed while staging (dyr, zhat as above),
 *   dX[E,cin] = dZ * W (gradient w.r.t. this layer's input activation; NULL = not needed),
 *   dW[C,cin] = dZ^T * act(Aprev) with act = the previous layer's BatchNorm+ReLU applied on the fly
 *   to its raw output Aprev (pscale == NULL: Aprev is the plain input), and
 *   psums[0:cin] += sum dxr, psums[cin:2cin] += sum dxr*zhat_prev: the BatchNorm-backward sums of
 *   the PREVIOUS layer, so that no separate reduce pass is needed for it.
 *   Wb = W (layout [C][cin]) packed tile-major [ceil(cin/32)][round4(C)][32];
 *   Wg = the same W packed in column blocks of 4/2/1 tiles of 32 input channels, each block
 *        [round4(C)][32][nt] (nt tiles of the block interleaved so that one vector load per k feeds
 *        nt MFMAs).  Wg != NULL selects the split schedule (one dX kernel + one dW kernel); NULL
 *        the single-kernel schedule.  gridgcn_pack_linear writes every layout in one launch. */
int gridgcn_pack_linear(const float *W, const float *b, int C, int cin_w, int rot, int cin, int ndx,
                        float *Wp, float *Bp, float *Wb, float *Wg, float *Wq, float *Wdx,
                        void *stream);
/* The same for MANY layers in one launch (the weights of a network change once per optimizer step).
 * The caller keeps an array of descriptors in DEVICE memory: fill each one on the host (pointers and
 * C, cin_w, rot, cin, ndx as for gridgcn_pack_linear; NULL for a layout that is not wanted), let
 * gridgcn_pack_desc_fill() complete K / ldw / n, copy the array to the device once, and call
 * gridgcn_pack_linear_batch(device array, layers, max over the layers' n) every step. */
typedef struct gridgcn_pack_desc {
    const float *W, *b;
    float *Wp, *Bp, *Wb, *Wg, *Wq, *Wdx;
    float *wgb;                 /* optional [4][C]: rows 0..2 = W[:, 0:3]^T (geo != 0) or zeros, row 3 = b:
                                 * the geo_vec weights + bias table of the gridgcn_edge_lin0_* kernels */
    int32_t C, cin_w, rot, cin, ndx;
    int32_t K, ldw, n;          /* written by gridgcn_pack_desc_fill */
    int32_t geo, reserved;
} gridgcn_pack_desc;
int gridgcn_pack_desc_fill(gridgcn_pack_desc *desc_host);
int gridgcn_pack_linear_batch(const gridgcn_pack_desc *descs_dev, int nlayers, int max_n, void *stream);
/* gridgcn_linear_fwd_direct: as gridgcn_linear_fwd for X[E][K] with K % 8 == 0 (zero-padded input
 *   channels), weights in the "Wq" order: each lane reads 16 consecutive floats of its row straight
 *   into registers and consumes them over 16 MFMA steps -- step (c, q, i) of lane l multiplies
 *   k = 32c + (l>>5)*4*nq + 4q + i (nq = 4, or K%32/8 in the last chunk) -- so Wq is
 *   [K/2 steps][64 lanes][ldw/32] with element (s, l, t) = W[t*32 + (l&31)][k(s, l)].  No LDS
 *   staging of X, 16 waves per CU. */
int gridgcn_linear_fwd_direct(const float *X, long long E, int K, int ldx, const float *Wq,
                              const float *b, int ldw, int cout, const float *scale,
                              const float *shift, float *Z, double *sums, void *stream);
int gridgcn_linear_fwd_direct_ld(const float *X, long long E, int K, int ldx, const float *Wq,
                                 const float *b, int ldw, int cout, const float *scale,
                                 const float *shift, void *Z, double *sums, int ldz, int zfmt,
                                 void *stream);
/* Dropout without a dropped tensor (the class-score conv of the segmentation head, ggcn_models_g.py:36-38:
 * fc1 -> Dropout -> fc2).  gridgcn_linear_fwd_direct_drop: Z = dropout(relu(X * scale + shift)) * W + b with the
 * mask evaluated while the rows are loaded (cout <= 32, K % 32 == 0, fp32 mode); gridgcn_linear_dw_drop:
 * dW[C][cin] = dZ^T dropout(relu(Aprev * pscale + pshift)) with dZ formed from (dY, Z, scale .. m2) as in
 * gridgcn_linear_bwd (cin == 128, C <= 32; workspace: gridgcn_linear_bwd_workspace_bytes).  The mask of element
 * (row, column) is the hash of row * K + column and (drop_seed + *drop_seed_dev) that
 * gridgcn_bn_relu_dropout_apply and gridgcn_linear_dx evaluate: the three agree bit for bit.  Other shapes:
 * GRIDGCN_EINVAL (use gridgcn_bn_relu_dropout_apply + the plain entries). */
int gridgcn_linear_fwd_direct_drop(const float *X, long long E, int K, int ldx, const float *Wq,
                                   const float *b, int ldw, int cout, const float *scale,
                                   const float *shift, float *Z, float drop_p, uint64_t drop_seed,
                                   const uint64_t *drop_seed_dev, void *stream);
int gridgcn_linear_dw_drop(const float *dY, const float *Z, const float *scale, const float *shift,
                           const float *mean, const float *rstd, const float *m1, const float *m2,
                           const float *Aprev, const float *pscale, const float *pshift, long long E, int C,
                           int cin, float drop_p, uint64_t drop_seed, const uint64_t *drop_seed_dev,
                           float *dW, void *workspace, size_t workspace_bytes, void *stream);
/* gridgcn_linear_fwd_direct_fin: the same launch also FINALISES the layer's BatchNorm -- what
 *   gridgcn_bn_finalize[_tail] does in a launch of its own (scale / shift / mean / rstd of the batch
 *   statistics, running statistics, num_batches_tracked) is done by the last workgroup to arrive
 *   (a few hundred persistent workgroups: one relaxed ticket each; ~26 launches less per training step).
 *   fin->ticket: one int32, zero at the call.  sums must be given. */
typedef struct gridgcn_bn_fin {
    const float *gamma, *beta;
    float *scale, *shift, *mean, *rstd;     /* [cout + tail] tables (outputs) */
    float *running_mean, *running_var;      /* may be NULL */
    int64_t *num_batches_tracked;           /* may be NULL */
    int32_t *ticket;
    float eps, momentum;
    int32_t tail, reserved;
} gridgcn_bn_fin;
int gridgcn_linear_fwd_direct_fin(const float *X, long long E, int K, int ldx, const float *Wq,
                                  const float *b, int ldw, int cout, const float *scale,
                                  const float *shift, void *Z, double *sums, int ldz, int zfmt,
                                  const gridgcn_bn_fin *fin, void *stream);
/* (zfmt 1: Z is written as bf16, round to nearest even, ldz in elements -- "bf16 storage" of a large
 *  per-edge pre-activation in the bf16 mode of BASELINE configs[2]; the BatchNorm statistics are those
 *  of the fp32 values.  Its readers: gridgcn_pairmax_fwd_src_z (za_bf16) and gridgcn_linear_bwd_ld
 *  (zfmt 1, shapes of the fused attention backward only).) */
/* (ldx = row stride of X in floats, >= K, a multiple of 4, X 16-byte aligned; sums may be NULL;
 *  Z may be NULL when sums is given: a statistics-only pass that stores nothing.) */
/* ---- classification edge block (classification/models/gcn_module_g.py:64-114 verts_pair_func
 *      with att_full='next'; :212-223 contextvec_func) -----------------------------------------
 * The classifier's attention MLP reads concat(att1(att_vec), pt_mlp(nf), context) per edge; the
 * concatenation is never written:
 * gridgcn_linear_fwd_direct2: as gridgcn_linear_fwd_direct with the K1 + K2 input columns taken from
 *   two tensors (X1[E][ld1] columns [0,K1), K1 % 32 == 0; X2[E][ld2] columns [0,K2), K2 % 8 == 0;
 *   Wq / scale / shift indexed by the concatenated column) and, when rowbias != NULL, a bias per
 *   group of P consecutive rows rowbias[E/P][cout] instead of b (P % 32 == 0): the context term
 *   ctx[centre] Wc^T + b, constant over a centre's neighbours.
 * gridgcn_ctx_max: ctx[B*O][3 + Cf] = max over the P neighbours of (xyz_nbr - xyz_centre | features
 *   of the neighbour) read from src[B][Nsrc][Cs = 4 + Cf] through nebidx (take mode 'clip');
 *   cidx[B*O][Cf] (may be NULL) = flat source row of the arg-max of every feature column.
 * gridgcn_ctx_max_backward: dsrc[cidx[c][j]][4 + j] += dctx[c][3 + j] (atomic; dsrc [B*Nsrc][Cs]).
 * gridgcn_bn_dz_segsum: out[c][:] = sum over the P rows of centre c of dZ, dZ = the BatchNorm+ReLU
 *   backward of (dY, Z) as in gridgcn_linear_bwd: the gradient of the per-centre bias.
 * gridgcn_sparse_add: dX[c*P + amax[c][ch]][ch] += gval[c][ch]: the sparse product/max gradient
 *   (gridgcn_pairmax_bwd) folded into a dense gradient of the same tensor. */
int gridgcn_linear_fwd_direct2(const float *X1, int ld1, int K1, const float *X2, int ld2, int K2,
                               long long E, const float *Wq, const float *b, const float *rowbias,
                               int P, int ldw, int cout, const float *scale, const float *shift,
                               float *Z, double *sums, void *stream);
int gridgcn_ctx_max(const float *src, const int32_t *nebidx, const float *cent, int cent_stride,
                    int B, int Nsrc, int Cs, int O, int P, float *ctx, int32_t *cidx, void *stream);
int gridgcn_ctx_max_backward(const float *dctx, const int32_t *cidx, long long ncent, int Cf, int Cs,
                             float *dsrc, void *stream);
int gridgcn_bn_dz_segsum(const float *dY, const float *Z, const float *scale, const float *shift,
                         const float *mean, const float *rstd, const float *m1, const float *m2,
                         long long ncent, int P, int C, float *out, void *stream);
/* gridgcn_gemm_small: the small dense products beside the edge pipeline (the first point conv applied
 *   to the source points and its two backward products; csrc/gridgcn_gemm.hip), fp32 MFMA, operands
 *   with arbitrary row strides (column slices of wider tensors), no packing:
 *     mode 0  C[M][N] = A[M][K] * B[N][K]^T        (any K since ABI 5)
 *     mode 1  C[M][N] = A[M][K] * B[K][N]          (any K; zero_left <= 32: the columns
 *                                                   [-zero_left, 0) left of C are zero-filled)
 *     mode 2  C[M][N] = A[K][M]^T * B[K][N]        (contraction over the K rows, any K; fixed summation
 *                                                   order; workspace of _workspace_bytes whose last
 *                                                   tiles*4 + 256 bytes are ZERO at the first call --
 *                                                   the kernel leaves them zero, so the buffer can be
 *                                                   reused by later calls of the same shape) */
int gridgcn_gemm_small_workspace_bytes(int M, int N, int K, size_t *bytes);
int gridgcn_gemm_small(int mode, const float *A, int lda, const float *B, int ldb, float *C, int ldc,
                       int M, int N, int K, int zero_left, void *workspace, size_t workspace_bytes,
                       void *stream);
/* gridgcn_gemm_bias: modes 0 / 1 with bias[N] (may be NULL) added to every row: torch's addmm / Linear for the
 * products beside the edge pipeline that no conv+BatchNorm kernel takes (per-centre context bias of the
 * classification block, layers of a handful of rows). */
int gridgcn_gemm_bias(int mode, const float *A, int lda, const float *B, int ldb, const float *bias, float *C,
                      int ldc, int M, int N, int K, void *stream);
/* gridgcn_bn_stats: sums[c] += sum_e Z[e][c], sums[C+c] += sum_e Z[e][c]^2 (input of
 *   gridgcn_bn_finalize) for a layer whose GEMM ran elsewhere (the "wide" fallback: stacks beyond
 *   the MFMA kernels' 256 output / 384 input channels use gridgcn_gemm_small / _bias + these BatchNorm
 *   kernels). */
int gridgcn_bn_stats(const float *Z, long long E, int C, int ld, double *sums, void *stream);
int gridgcn_sparse_add(const uint8_t *amax, const float *gval, long long ncent, int P, int C,
                       float *dX, void *stream);
/* gridgcn_pack_linear: W[C][cin_w] (framework layout, C <= 256), b[C]; the kernels see `cin` >=
 *   cin_w input channels: kernel column k = framework column k + rot (k < cin_w - rot), k - (cin_w -
 *   rot) (k < cin_w), zero (k >= cin_w) -- i.e. the first `rot` columns moved behind the others and
 *   zero padding, the row layout of gridgcn_edge_inputs_rows (rot = 3: geo_vec).  Outputs:
 *   Wdx[round8(C)/2 * 64 * ntv] (ndx > 0): dX operand of gridgcn_linear_bwd's direct schedule for the
 *   first ndx input channels, ntv = ceil(ndx/32) rounded up to 1/2/4/8, channel order as Wq;
 *   Wp[round4(cin) * ldw] / Bp[ldw] for gridgcn_linear_fwd (ldw = C rounded up to 32/64/128/256),
 *   Wq[round8(cin) * ldw] for gridgcn_linear_fwd_direct,
 *   Wb, Wg [ceil(cin/32) * round4(C) * 32] each for gridgcn_linear_bwd.  Any output may be NULL.
 * gridgcn_bn_finalize: batch statistics -> scale = gamma*rstd, shift = beta - mean*scale, mean,
 *   rstd = rsqrt(var_biased + eps) from sums = (sum z, sum z^2) over E rows; running_mean/var
 *   (NULL = not tracked) updated with `momentum`, the variance unbiased (mx.sym.BatchNorm /
 *   utils/ops.py:149-158 semantics, as torch.nn.BatchNorm1d).
 * gridgcn_bn_bwd_finalize: sums = (s1, s2) of gridgcn_bn_relu_bwd_reduce -> m1 = s1/E, m2 = s2/E,
 *   dbeta = s1, dgamma = s2. */
int gridgcn_bn_finalize(const double *sums, const float *gamma, const float *beta, long long E,
                        float eps, float momentum, int C, float *scale, float *shift, float *mean,
                        float *rstd, float *running_mean, float *running_var,
                        int64_t *num_batches_tracked, void *stream);
/* gridgcn_bn_finalize_tail: scale .. rstd are rows of a table of C + tail columns; the tail columns get
 * the identity (scale 1, shift / mean / rstd 0): the concat(centre MLP output, aggregate) buffer whose
 * consumer applies the producer's BatchNorm while loading (gridgcn_linear_bwd_ld, nbn). */
int gridgcn_bn_finalize_tail(const double *sums, const float *gamma, const float *beta, long long E,
                             float eps, float momentum, int C, int tail, float *scale, float *shift,
                             float *mean, float *rstd, float *running_mean, float *running_var,
                             int64_t *num_batches_tracked, void *stream);
int gridgcn_bn_bwd_finalize(const double *sums, long long E, int C, float *m1, float *m2,
                            float *dgamma, float *dbeta, void *stream);
int gridgcn_linear_bwd_workspace_bytes(long long E, int cin, int C, size_t *bytes);
int gridgcn_linear_bwd(const float *dY, const float *Z, const float *scale, const float *shift,
                       const float *mean, const float *rstd, const float *m1, const float *m2,
                       const float *Aprev, const float *pscale, const float *pshift,
                       const float *pmean, const float *prstd, const float *Wb, const float *Wg,
                       const float *Wdx, int ndx, long long E,
                       int C, int cin, int cin_w, int rot, int ldy, float *dX, float *dW,
                       double *psums,
                       const uint8_t *amax, const float *gval, int P, void *workspace,
                       size_t workspace_bytes, void *stream);
int gridgcn_linear_bwd_ld(const float *dY, const void *Z, const float *scale, const float *shift,
                          const float *mean, const float *rstd, const float *m1, const float *m2,
                          const float *Aprev, const float *pscale, const float *pshift,
                          const float *pmean, const float *prstd, const float *Wb, const float *Wg,
                          const float *Wdx, int ndx, long long E,
                          int C, int cin, int cin_w, int rot, int ldy, int ldz, int nbn, int zfmt,
                          float *dX, float *dW, double *psums,
                          const uint8_t *amax, const float *gval, int P, void *workspace,
                          size_t workspace_bytes, void *stream);
/* gridgcn_linear_bwd_fin = gridgcn_bn_bwd_finalize + gridgcn_linear_bwd_ld without the finalisation
 *   launch: `sums` [2][C] fp64 (s1, s2) as gridgcn_bn_relu_bwd_reduce or a dX epilogue (psums) left
 *   them.  The register-direct kernels divide by E themselves; m1, m2, dgamma, dbeta [C] are OUTPUTS
 *   (written by the dW reduce kernel, or by gridgcn_bn_bwd_finalize's kernel when a legacy kernel
 *   is on the path). */
int gridgcn_linear_bwd_fin(const float *dY, const void *Z, const float *scale, const float *shift,
                           const float *mean, const float *rstd, const double *sums, float *m1, float *m2,
                           float *dgamma, float *dbeta,
                           const float *Aprev, const float *pscale, const float *pshift,
                           const float *pmean, const float *prstd, const float *Wb, const float *Wg,
                           const float *Wdx, int ndx, long long E,
                           int C, int cin, int cin_w, int rot, int ldy, int ldz, int nbn, int zfmt,
                           float *dX, float *dW, double *psums,
                           const uint8_t *amax, const float *gval, int P, void *workspace,
                           size_t workspace_bytes, void *stream);
/* (cin = row length of Aprev / dX as the kernels see it; dW is written in the FRAMEWORK layout
 *  [C][cin_w]: zero-padding columns dropped and the `rot` columns moved back in front, the inverse
 *  of gridgcn_pack_linear's mapping.  cin_w = cin, rot = 0 for an ordinary layer.) */
/* (Wdx != NULL: dX columns 0..ndx-1 only, by the register-direct schedule -- dZ is formed in
 *  registers from 16-byte row reads, no LDS staging; needs C % 8 == 0, falls back otherwise.)
 * (amax != NULL: the upstream gradient is the sparse one of gridgcn_pairmax_bwd -- row e belongs to
 *  centre e/P, neighbour e%P; dY[e,c] = (amax[e/P,c] == e%P) ? gval[e/P,c] : 0 -- and dY is ignored.)
 *
 * gridgcn_pairmax_fwd: agg[o,c] = max_p relu(Zp*scale_p+shift_p) * relu(Za*scale_a+shift_a) over the
 *   P rows of centre o (Zp, Za [ncent*P, C] pre-BatchNorm outputs of the last pt / att layer;
 *   gcn_module_g_att.py:167 and :57-59), amax = first arg max, ONE BYTE per (centre, channel)
 *   (P <= 256; the arg-max tensors are read three to five times per step).
 * gridgcn_pairmax_bwd: gp/ga[o,c] = gradient w.r.t. the two post-ReLU activations at the arg-max
 *   edge, and the BatchNorm-backward sums (as gridgcn_bn_relu_bwd_reduce) of both layers. */
int gridgcn_pairmax_fwd(const float *Zp, const float *Za, const float *scale_p,
                        const float *shift_p, const float *scale_a, const float *shift_a,
                        long long ncent, int P, int C, float *agg, int ld_agg, uint8_t *amax,
                        float *zsel, void *stream);
/* ld_agg / ldy (gridgcn_bn_relu_apply): row stride in floats of the output, >= C -- lets the two
 * halves of update_func's concat (gcn_module_g_att.py:31-36) be written in place, no concat pass. */
/* zsel (optional, [2][ncent*C]): the pre-BatchNorm values of Zp and Za at the arg max, written by
 * the forward and read by the backward instead of gathering them again from Zp / Za. */
int gridgcn_pairmax_bwd(const float *Zp, const float *Za, const float *scale_p,
                        const float *shift_p, const float *mean_p, const float *rstd_p,
                        const float *scale_a, const float *shift_a, const float *mean_a,
                        const float *rstd_a, const float *dagg, const uint8_t *amax,
                        long long ncent, int P, int C, int ld_dagg, float *gp, float *ga,
                        double *sums_p,
                        double *sums_a, const float *zsel, void *stream);
/* gridgcn_pairmax_bwd_masked: the same from zsel alone, with ga = the sparse term of the attention layer's dZ
 * itself: scale_a * (relu(bn(za)) > 0 ? gradient : 0) -- for a consumer that has no pre-activation to mask with
 * (gridgcn_att_bwd_noz). */
int gridgcn_pairmax_bwd_masked(const float *scale_p, const float *shift_p, const float *mean_p,
                               const float *rstd_p, const float *scale_a, const float *shift_a,
                               const float *mean_a, const float *rstd_a, const float *dagg,
                               const uint8_t *amax, long long ncent, int P, int C, int ld_dagg, float *gp,
                               float *ga, double *sums_p, double *sums_a, const float *zsel, void *stream);

/* ---- backward of the second attention conv WITHOUT its [E, C] pre-activation -----------------------------
 * (update_att_mlp2d_scnd, gcn_module_g_att.py:152: 32 -> 128 channels, behind the BatchNorm'd 10 -> 32 conv,
 * in front of the neighbour max pool.)  BatchNorm backward gives dZ2 = sc (mask ? g : 0) + (z2 - mu) bz + cz:
 * the first term is sparse (the arg-max edge per (centre, channel): amax, gval -- gval = that term itself, as
 * gridgcn_pairmax_bwd_masked leaves it), the second affine in z2 = W2 a1 + b2, a1 = relu(bn1(Z1)); so
 *   dA1 = dZ2_sparse W2 + a1 (W2^T diag(bz) W2) + W2^T (cz + bz (b2 - mu)),
 *   dW2 = dZ2_sparse^T a1 + (cz + bz (b2 - mu)) (sum_e a1) + diag(bz) W2 (sum_e a1 a1^T)
 * need Z1 [E, 32] and the sparse gradient only; Z2 [E, 128] is neither read nor kept for the backward.
 * in : Z1[E,32] with its BatchNorm vectors p* [32]; W2[128,32], b2[128] (framework layout); this layer's scale
 *      (= gamma * rstd), mean, rstd [128]; sums [2][128] (fp64) = its BatchNorm-backward sums (pairmax_bwd);
 *      amax / gval [E / P, 128].
 * out: dX[E,32] (gradient w.r.t. relu(bn1(Z1))), dW[128,32], m1 / m2 / dgamma / dbeta [128] of this layer,
 *      psums [2][32] += BatchNorm-backward sums of the layer in front, s1 [32] += sum_e a1 (fp64, both zeroed
 *      by the caller).  cin == 32, C == 128, E % P == 0, E >= 32; other shapes: GRIDGCN_EINVAL (use
 *      gridgcn_linear_bwd, which reads Z2). */
int gridgcn_att_bwd_noz_workspace_bytes(long long E, int cin, int C, size_t *bytes);
int gridgcn_att_bwd_noz(const float *Z1, const float *pscale, const float *pshift, const float *pmean,
                        const float *prstd, const float *W2, const float *b2, const float *scale,
                        const float *mean, const float *rstd, const double *sums, const uint8_t *amax,
                        const float *gval, int P, long long E, int cin, int C, float *dX, float *dW, float *m1,
                        float *m2, float *dgamma, float *dbeta, double *psums, double *s1, void *workspace,
                        size_t workspace_bytes, void *stream);
/* gridgcn_att_bwd_noz_mom (round 6): the same backward with S1 = sum_e a1 and S2 = sum_e a1 a1^T TAKEN from the
 * moments the Z2-free forward of the layer left behind (gridgcn_att_bn2_moments: fp64, at byte
 * gridgcn_att_moments_offset(E, ..) of ITS workspace -- 17 x 64 doubles: rows 0..15 the C/D registers of S2, lane l =
 * column l & 31, row (r & 3) + 8 (r >> 2) + 4 (l >> 5); row 16 lane l's share of S1[l & 31]) instead of being
 * accumulated again: 144 MFMAs per 32-row tile instead of 160, reduce + finish in one launch.  Same outputs as
 * gridgcn_att_bwd_noz except `s1` (not produced); dW's dense part is evaluated in fp64 from the fp64 moments.  The
 * caller keeps the forward's workspace alive until here.  _supported: 1 when the shape is taken (the limits of
 * gridgcn_att_bwd_noz and E < 2^24, E / P < 2^22), else 0 and the call returns GRIDGCN_EINVAL. */
int gridgcn_att_bwd_noz_mom_supported(long long E, int cin, int C, int P);
int gridgcn_att_moments_offset(long long E, int cin, int C, size_t *offset_bytes);
int gridgcn_att_bwd_noz_mom(const float *Z1, const float *pscale, const float *pshift, const float *pmean,
                            const float *prstd, const float *W2, const float *b2, const float *scale,
                            const float *mean, const float *rstd, const double *sums, const uint8_t *amax,
                            const float *gval, int P, long long E, int cin, int C, const double *moments, float *dX,
                            float *dW, float *m1, float *m2, float *dgamma, float *dbeta, double *psums,
                            void *workspace, size_t workspace_bytes, void *stream);
/* ---- FORWARD of the same pair of layers without that pre-activation (csrc/gridgcn_attfwd.hip) -------------------
 * Training needed Z2 = W2 a1 + b2 [E, 128] for two things: its BatchNorm batch statistics and the pair product /
 * neighbour max (gcn_module_g_att.py:152-167, :57-59).  Both are obtained from Z1 [E, 32]:
 * gridgcn_att_bn2_moments: S1 = sum_e a1, S2 = sum_e a1 a1^T (a1 = relu(Z1 * scale1 + shift1); fp32 MFMA products,
 *   fp64 accumulation from 256 rows up, fixed summation order), then for every output channel
 *     sum z2 = W2[c,:].S1 + E b2[c],   sum z2^2 = W2[c,:] S2 W2[c,:]^T + 2 b2[c] W2[c,:].S1 + E b2[c]^2   (fp64)
 *   and from those exactly what gridgcn_bn_finalize writes (scale, shift, mean, rstd [128]; the running estimates
 *   and num_batches_tracked when given; sums [2][128] fp64 when not NULL).  cin = 32, C = 128.
 * gridgcn_att_pairmax_fwd: what gridgcn_pairmax_fwd_src writes (agg, amax, zsel -- zsel REQUIRED: the backward has
 *   nothing else to take the arg-max pre-activations from), the second conv recomputed per 30-edge tile on the MFMA
 *   unit.  P = 5, cin = 32, C = 128, O >= 6, ncent >= 7, B*Nsrc < 2^23, ncent*ld_agg < 2^30; GRIDGCN_EINVAL otherwise (callers keep the Z2 path).
 *   The attention value of an edge is W2 a1 + b2 in the MFMA unit's summation order with the bias FIRST: the same
 *   terms as gridgcn_linear_fwd_direct, last-bit differences possible; the point branch is bit-identical. */
int gridgcn_att_fwd_noz_workspace_bytes(long long E, int cin, int C, size_t *bytes);
/* 1 when gridgcn_att_pairmax_fwd takes this shape (rows = B*Nsrc), 0 when it would return GRIDGCN_EINVAL: the ONE
 * statement of its limits -- callers decide between the Z2-free pair and the Z2 path with it, before anything runs. */
int gridgcn_att_pairmax_fwd_supported(long long ncent, int O, int P, int cin, int C, int ld_agg, long long rows);
int gridgcn_att_bn2_moments(const float *Z1, const float *scale1, const float *shift1, const float *W2,
                            const float *b2, const float *gamma, const float *beta, long long E, int cin, int C,
                            float eps, float momentum, float *scale, float *shift, float *mean, float *rstd,
                            float *running_mean, float *running_var, int64_t *num_batches_tracked, double *sums,
                            void *workspace, size_t workspace_bytes, void *stream);
int gridgcn_att_pairmax_fwd(const float *Ysrc, const int32_t *nebidx, const float *att16, const float *Wg,
                            const float *b, int B, int Nsrc, int O, const float *Z1, const float *scale1,
                            const float *shift1, const float *W2, const float *b2, const float *scale_p,
                            const float *shift_p, const float *scale_a, const float *shift_a, long long ncent,
                            int P, int cin, int C, float *agg, int ld_agg, uint8_t *amax, float *zsel, void *stream);
int gridgcn_bn_relu_apply(const float *Z, const float *scale, const float *shift, float *Y,
                          long long E, int C, int ldy, void *stream);
/* Head of the segmentation net: fc1 (conv+BN+ReLU) -> Dropout(p) -> fc2
 * (segmentation/models/ggcn_models_g.py:33-38).  The Dropout is folded into the two kernels either
 * side of it; its mask is a counter-based hash of (drop_seed, element index row*C + col) that the
 * backward regenerates, so no mask is stored and no separate pass runs:
 *   gridgcn_bn_relu_dropout_apply: Y = dropout(relu(Z*scale + shift)), kept values * 1/(1-p).
 *   gridgcn_linear_dx: the dX half of gridgcn_linear_bwd on its own (register-direct schedule only,
 *     EINVAL if the shape is outside it): dX[E,0:ndx] = dZ W, times the dropout mask of the
 *     [E][cin] input activation when drop_p > 0 (ndx = cin then), and psums of the layer in front
 *     (Aprev = its raw output, pscale.. its BatchNorm) accumulated from the masked dX.
 * 0 <= drop_p < 1; drop_p = 0 is the identity. */
int gridgcn_bn_relu_dropout_apply(const float *Z, const float *scale, const float *shift, float *Y,
                                  long long E, int C, int ldy, float drop_p, uint64_t drop_seed,
                                  const uint64_t *drop_seed_dev,
                                  void *stream);
int gridgcn_linear_dx(const float *dY, const float *Z, const float *scale, const float *shift,
                      const float *mean, const float *rstd, const float *m1, const float *m2,
                      const float *Aprev, const float *pscale, const float *pshift,
                      const float *pmean, const float *prstd, const float *Wdx, int ndx,
                      long long E, int C, int cin, int ldy, float drop_p, uint64_t drop_seed,
                      const uint64_t *drop_seed_dev,
                      float *dX, double *psums, void *stream);
int gridgcn_bn_relu_bwd_reduce(const float *dY, const float *Z, const float *scale,
                               const float *shift, const float *mean, const float *rstd,
                               long long E, int C, int ldy, double *sums, void *stream);
/* ldy (also gridgcn_linear_bwd, dense dY only) / ld_dagg: row stride of the upstream gradient, so
 * that the two halves of update_func's concat gradient are consumed in place. */
int gridgcn_bn_relu_bwd_elemt(const float *dY, const float *Z, const float *scale,
                              const float *shift, const float *mean, const float *rstd,
                              const float *m1, const float *m2, long long E, int C, float *dZ,
                              void *stream);

/* ---- segmentation head loss: SoftmaxOutput(use_ignore, ignore_label, normalization='valid')
 *      (segmentation/models/ggcn_models_g.py:41) ------------------------------------------------
 * logits[E][ld] f32 rows zero-padded to ld floats (ld a multiple of 4, <= 32; ncls <= ld classes),
 * label[E] i64.  fwd: lse[E] = log-sum-exp per row; acc[0] += sum over counted rows of
 * -log softmax[label], acc[1] += number of counted rows (label != ignore_label; acc zeroed by the
 * caller; loss = acc[0]/acc[1]).  bwd: dlogits[E][ld] = (softmax - onehot) * grad_loss[0] / acc[1]
 * in counted rows, 0 elsewhere and in the padding columns.
 * gridgcn_colsum: out[c] += sum_e X[e][c], c < ncols (X rows of ld floats as above). */
int gridgcn_softmax_ce_fwd(const float *logits, int ld, int ncls, const int64_t *label, long long E,
                           int ignore_label, float *lse, double *acc, void *stream);
int gridgcn_softmax_ce_bwd(const float *logits, int ld, int ncls, const int64_t *label, long long E,
                           int ignore_label, const float *lse, const double *acc,
                           const float *grad_loss, const float *class_weight, float *dlogits,
                           void *stream);
/* class_weight (optional, [ncls]): the reference's 'weighted_gradient' custom op between fc2 and
 * the loss (custom_op/weighted_gradient.py:18-26, ggcn_models_g.py:40): every row of dlogits is
 * multiplied by max_c [dlogits_c < 0] * class_weight_c, i.e. by the weight of the row's label. */
int gridgcn_colsum(const float *X, long long E, int ld, int ncols, double *out, void *stream);
/* gridgcn_softmax_ce_loss: gridgcn_softmax_ce_fwd with the two sums spread over 16 slots (atomics on ONE
 *   address are served one after the other) and loss[0] = sum / max(count, 1) written by the last
 *   workgroup to arrive.  acc = fp64[544], zeroed by the caller, 128-byte aligned: 16 slots x 16, then
 *   acc[256] = sum, acc[257] = count (pass acc + 256 to gridgcn_softmax_ce_bwd), then 17 ticket lines.
 * gridgcn_colsum_f32: gridgcn_colsum likewise, out[c] = (float) total; acc fp64[784], zeroed, 128-byte
 *   aligned (16 slots x 32 partial sums, then 17 ticket lines). */
int gridgcn_softmax_ce_loss(const float *logits, int ld, int ncls, const int64_t *label, long long E,
                            int ignore_label, float *lse, double *acc3, float *loss, void *stream);
int gridgcn_colsum_f32(const float *X, long long E, int ld, int ncols, double *acc, float *out,
                       void *stream);

/* ---- glue of a layer boundary and the optimizer -------------------------------------------------
 * gridgcn_cat_mask: out[r, 0:ca] = a[r, :], out[r, ca:ca+cb] = b[r, :] * mask[r], zeros up to ldo;
 *   b == NULL: the cb columns are 1.0; mask == NULL: no mask; out2 (optional, row stride ldo2): the
 *   same rows once more (the zero-padded copy).  data_layer = concat(cent, feats * centmsk):
 *   segmentation/models/ggcn_models_g.py:137,186,231, gcn_module_g_att.py:284-285.
 * gridgcn_mask_sum: out[r, c] = (g1[r, col0 + c] + g2[r, col0 + c]) * mask[r], c < C: its backward
 *   (g1 / g2 with row strides ld1 / ld2; either may be NULL).
 * gridgcn_adam_step: the Adam update of n tensors in one launch per 128 tensors (base_solver.py:105-114:
 *   mx.optimizer.Adam with wd).  params / grads / sizes / mchunk are HOST arrays (device pointers,
 *   element counts, first 1024-element chunk of tensor i in the moment buffers m and v); state =
 *   int32[2] on the device (step count t, ticket), zero before the first call, t += 1 per call;
 *   lr_dev (optional): learning rate read from device memory instead of lr.
 *   mode 0: torch.optim.Adam  w -= lr/(1-b1^t) m / (sqrt(v)/sqrt(1-b2^t) + eps);
 *   mode 1: mx.optimizer.Adam w -= lr sqrt(1-b2^t)/(1-b1^t) m / (sqrt(v) + eps); g = grad + wd w. */
int gridgcn_cat_mask(const float *a, int lda, int ca, const float *b, int ldb, int cb,
                     const float *mask, float *out, int ldo, float *out2, int ldo2, long long E,
                     void *stream);
int gridgcn_mask_sum(const float *g1, int ld1, const float *g2, int ld2, int col0, int C,
                     const float *mask, float *out, long long E, void *stream);
int gridgcn_adam_step(float *const *params, const float *const *grads, const long long *sizes,
                      const long long *mchunk, int n, float *m, float *v, int32_t *state, float lr,
                      const float *lr_dev, float beta1, float beta2, float eps, float weight_decay,
                      int mode, void *stream);

/* ---- GridConv edge pipeline (inference-mode BatchNorm) ----------------------------------------
 * Replaces, for one sub_g_update call (segmentation/models/gcn_module_g_att.py:172-287, aggtype
 * 'gcn', attfdim 10, pool max), the operators between the index op and update_func:
 *   batch_take_g (utils/ops.py:78-93) -> geo_vec/geo_dist/att_vec (:190-194,217-218) ->
 *   [concat geo_vec if localfdim != 0] (:242-250) -> pt-MLP (:135) -> att-MLP (:141,152) ->
 *   product (:167) -> max over P (:57-59).
 * src[B,Nsrc,Cs] f32 (x,y,z,w,features; Cs = 4 + C_in), nebidx[B,O,P] i32 (take mode 'clip'),
 * cent: xyz of centre ci at cent + ci*cent_stride floats, out[B,O,C] f32.
 * Each 1x1 conv is given BatchNorm-folded and zero padded: contraction length K (multiple of 4),
 * width ldw = cout rounded up to 32/64/128/256, bias b[ldw], and W packed per group of up to 128
 * output columns as [group][K][32][NT] (NT = group columns / 32; element (k, j, t) is the weight
 * of input row k and output column group*128 + t*32 + j).  Input rows of the first pt layer are
 * the columns of the gathered source row with columns 0..3 replaced by (geo_vec, 0), i.e.
 * K = round4(Cs); of the first att layer the 10 att_vec channels (K = 12); of every other layer
 * the previous layer's outputs (K = round4(cout_prev)).  pt: npt layers (1..4), intermediate
 * widths <= 128; att: exactly 2 layers (10 -> C/4 -> C).  Layer structs live in HOST memory,
 * W/b on the device.  grid_gcn_amd.ops.pack_conv_layer builds this layout. */
typedef struct gridgcn_conv_layer {
    const float *W;
    const float *b;
    int32_t K;
    int32_t ldw;
    int32_t cout;
    int32_t reserved;
} gridgcn_conv_layer;

int gridgcn_gridconv_forward(const float *src, const int32_t *nebidx, const float *cent,
                             int cent_stride, int B, int Nsrc, int Cs, int O, int P,
                             int has_feats, int localfdim, int npt, const gridgcn_conv_layer *pt,
                             const gridgcn_conv_layer *att, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GRIDGCN_H_ */
