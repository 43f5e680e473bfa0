// gridgcn_edgelin.h -- parameter blocks of gridgcn_edgelin.hip
#pragma once
#include <hip/hip_runtime.h>

struct GGEdgeLin0 {
    const float *Ysrc;   // [B*Nsrc][C0] or nullptr (layer without neighbour features)
    const float *src;    // [B*Nsrc][Cs]  x,y,z first
    const int *nebidx;   // [E]
    const float *cent;   // centre ci at cent + ci*cent_stride
    const float *Wg;     // [3][C0] geo_vec weights or nullptr
    const float *b;      // [C0]
    float *Z;            // [E][C0] (nullptr: statistics and att16 only)
    float *att16;        // [E][16]
    double *sums;        // [2][C0]
    int cent_stride, B, Nsrc, Cs, O, P, C0, E;
};

struct GGEdgeLin0Bwd {
    const float *Z;       // [E][C0], or nullptr: recomputed from (Ysrc, Wg, b) as in the forward
    const float *Ysrc, *Wg, *b;
    const float *dY;      // dense upstream gradient [E][C0] (nullptr: sparse)
    const unsigned char *amax;   // sparse: [B*O][C0] arg-max neighbour (one byte), value
    const float *gval;
    const float *scale, *shift, *mean, *rstd, *m1, *m2;   // [C0]
    const float *att16;   // [E][16] (geo_vec at columns 1..3)
    const int *index;     // nebidx [B][M]
    const int *perm, *keys, *rowptr;
    float *dYsrc;         // [B*N][C0], zero-filled
    double *dWg;          // [3][C0], zero-filled (nullptr: no geo term)
    int B, N, O, P, C0, M, cpc, chunk;   // cpc chunks of `chunk` sorted edges per cloud, one wave each
};

size_t gg_edge_lin0_sparse_workspace(int B, int N, int C);
int gg_edge_lin0_bwd_sparse(const int *nebidx, const float *att16, const unsigned char *amax,
                            const float *gval, const float *zsel, const float *Ysrc,
                            const float *Wg, const float *bias, const float *scale,
                            const float *shift, const float *mean, const float *rstd,
                            const float *m1, const float *m2, int B, int N, int O, int P, int C,
                            float *dYsrc, float *Gsum, double *wgs, double *gg, void *workspace,
                            hipStream_t st, int geo_given = 0);
int gg_edge_lin0_dwg(const double *wgs, const double *gg, const float *T, const float *wgb,
                     const float *scale, const float *mean, const float *rstd, const float *m1,
                     const float *m2, int C, float *dW, int ld, hipStream_t st);
int gg_edge_lin0_fwd(const GGEdgeLin0 &p, hipStream_t st);
size_t gg_edge_geo_workspace(int B, int N, long long edges_per_cloud);
int gg_edge_geo_fwd(const float *Ysrc, const float *src, const int *nebidx, const float *cent, int cent_stride,
                    int B, int N, int Cs, int O, int P, int C, const float *Wg, const float *bias, float *att16,
                    float *Gsum, double *gg, double *sums, void *workspace, hipStream_t st);
int gg_edge_lin0_bwd(GGEdgeLin0Bwd p, void *workspace, hipStream_t st);   // workspace: gg_csr_workspace
