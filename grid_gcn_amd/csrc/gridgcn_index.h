// gridgcn_index.h -- workspace layout + host entry of the shared voxel-index build.
#pragma once
#include "gridgcn_dev.h"
#include <stddef.h>

// Byte offsets into the caller's workspace (all 256-byte aligned).
struct GGIndexWs {
    size_t o_cnt;         // int [B*G]   voxel population            (coor_counter, gridify.cu:358)
    size_t o_slotfirst1;  // int [B*O]   first point id + 1 of the voxel in centre slot o
    size_t o_blkcnt;      // int [B*nblk] voxel leaders per 1024-point block
    size_t o_wsum;        // u64 [B*nblk] per-block sum |w| of in-grid points, bit 63 = non-integer seen
    size_t o_exact;       // int [B]     1: weights are small integers -> order-free total_weight
    size_t o_cursor;      // int [B]     bump allocator of segment space inside [b*N, (b+1)*N)
    size_t zero_bytes;    // the region [0, zero_bytes) is memset to 0 every call
    size_t o_off;         // int [B*G]   start of the voxel's segment in seg/sorted/bkt
    size_t o_vox;         // int [B*N]   voxel id of the point or -1 (dropped)
    size_t o_arr;         // int [B*N]   arrival order inside the voxel
    size_t o_seg;         // int [B*N]   point ids grouped by voxel, arrival order
    size_t o_sorted;      // int [B*N]   point ids grouped by voxel, ascending
    size_t o_bkt;         // int [B*N]   first P entries of a segment with population > P:
                          //             the S0 reservoir bucket (coor_to_pntidx, gridify.cu:357)
    size_t o_lead;        // u8  [B*N]   1 if the point is the first (smallest id) of its voxel
    size_t total;
    int nblk, nslab, S;
};

size_t gg_index_workspace_bytes(int B, int N, const GGGrid &gp, bool with_centres, GGIndexWs *ws);
int gg_index_build(const float *data, const int *np, int B, int N, const GGGrid &gp,
                   bool with_centres, int *centnum, char *wsbase, const GGIndexWs &w,
                   hipStream_t st);
int gg_index_init();
