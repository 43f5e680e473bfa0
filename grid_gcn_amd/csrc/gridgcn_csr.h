// gridgcn_csr.h -- edges of every cloud ordered by destination row (gridgcn_scatter.hip).
#pragma once
#include <hip/hip_runtime.h>

#define GG_CSR_PARTS 16
#define GG_CSR_CHUNK 256

// workspace: perm[B*M], keys[B*M], rowptr[B][N+3], hist[B][PARTS][N+2]  (int32)
size_t gg_csr_workspace(int B, int N, int M);
// counting sort of index[B][M] by key = clip(index + b*N) - (b*N - 1) in [0, N] (N+1: clipped into
// another cloud).  perm = edge ids in sorted order, keys = their keys, rowptr = exclusive offsets.
// returns 1 when N does not fit the LDS histogram.
int gg_csr_build(const int *index, int B, int N, int M, void *workspace, int **perm, int **keys,
                 int **rowptr, hipStream_t st);
