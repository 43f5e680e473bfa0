// gridgcn_index.h -- workspace layout + host entry of the shared voxel-index build.
#pragma once
#include "gridgcn_dev.h"
#include <stddef.h>

#define GG_CHUNK_MAX 4096  // points per chunk of the first split: 1024, 2048 or 4096
                           // (12-bit local point number)
#define GG_MAX_SLABS 1024  // slabs per cloud (per-wave counters of the first split live in LDS)
#define GG_MAX_SB 12       // log2 of the largest slab (voxels): per-wave counters of the second split
#define GG_XRB 4           // log2 of the voxel run that stays together in a slab (16 voxels = 128 B
                           // of the voxel table)
#define GG_MAX_CHUNKS 1024 // chunks per cloud (run table of a slab lives in LDS)

// Byte offsets into the caller's workspace (all 256-byte aligned).  What the query kernels read
// (cnt, off, sorted, bkt, slotfirst1, exact) means the same in both build generations.
struct GGIndexWs {
    // ---- read by the query kernels ----
    size_t o_vtab;        // int2 [B*G]  per voxel: .x = start of its segment in sorted/bkt (absolute),
                          //             .y = population (coor_counter, gridify.cu:358)
    size_t o_sorted;      // int [B*N]   point ids grouped by voxel, ascending
    size_t o_bkt;         // int [B*N]   first P entries of a segment with population > P:
                          //             the S0 reservoir bucket (coor_to_pntidx, gridify.cu:357)
    size_t o_slotfirst1;  // int [B*O]   first point id + 1 of the voxel in centre slot o
    size_t o_exact;       // int [B]     bit 0: weights are small integers -> order-free
                          //             total_weight; bit 1: every in-grid weight is exactly 1.0
    // ---- build scratch, two-level split (gridgcn_index.hip) ----
    size_t o_part;        // u32 [B*N]   chunk-local split by slab: (voxel number inside the slab) << 12
                          //             | local point number
    size_t o_ctab;        // int [B*nchunk*(nslab+1)] chunk-local exclusive slab offsets
    size_t o_lbm;         // u32 [B*ceil(N/32)] leader bitmap: bit i of cloud b = point i is the first
                          //             point of its voxel (zeroed by the first kernel of the call)
    size_t o_cursor;      // legacy: int [B] bump allocator of segment space
    size_t o_wsum;        // u64 [B*nblk] per-chunk sum |w| of in-grid points, bit 63 = non-integer
                          //             seen, bit 62 = weight != 1.0 seen
    // ---- build scratch, legacy generation (gridgcn_index_legacy.hip) ----
    size_t o_cnt, o_off;  // int [B*G] each (packed into o_vtab by a last kernel)
    size_t o_blkcnt, o_vox, o_arr, o_seg, o_leadflag;
    size_t zero_bytes;    // legacy: the region [0, zero_bytes) is memset to 0 every call
    size_t total;
    int nblk;             // legacy: 1024-point blocks; split: chunks per cloud
    int nslab, S;         // voxel slabs per cloud, voxels per slab
    int SB;               // split: log2(S)
    int KB, MB;           // split: log2(nslab), bits of the 16-voxel run number
    unsigned HA, HAinv;   // split: odd multiplier of the run-number hash and its inverse mod 2^MB
    int NW2;              // split: waves per workgroup of the slab kernel
    int CH;               // split: points per chunk
    int legacy;           // 1: built by the legacy generation
    int small;            // 1: built by the one-launch kernel for clouds <= 4096 points
};

size_t gg_index_workspace_bytes(int B, int N, const GGGrid &gp, bool with_centres, GGIndexWs *ws);
int gg_index_build(const float *data, const int *np, int B, int N, const GGGrid &gp,
                   bool with_centres, int *centnum, char *wsbase, const GGIndexWs &w,
                   hipStream_t st);
int gg_index_init();
// plan overrides for measurements: which = 0: shift of log2(slabs per cloud), 1: points per chunk
// (1024 / 2048 / 4096, 0 = automatic), 2: one-launch build for clouds <= 4096 points (1 = on)
void gg_index_set_tuning(int which, int value);
int gg_index_get_tuning(int which);

size_t gg_index_legacy_workspace_bytes(int B, int N, const GGGrid &gp, bool with_centres,
                                       GGIndexWs *ws);
int gg_index_legacy_build(const float *data, const int *np, int B, int N, const GGGrid &gp,
                          bool with_centres, int *centnum, char *wsbase, const GGIndexWs &w,
                          hipStream_t st);
int gg_index_legacy_init();

// gridgcn_cas.hip: coverage-aware refinement of the centre slots (parity unpinned, see there)
size_t gg_cas_workspace_bytes(int B, int N, const GGGrid &gp);
int gg_cas_refine(const float *data, const int *np, int B, int N, const GGGrid &gp, float beta,
                  int *slotfirst1, const int *centnum, const int2 *vtab, const int *sorted, char *ws,
                  hipStream_t st);

// gridgcn_fastrand.hip: centre slots + query of the fast_rand build (scratch: gg_cas_workspace_bytes)
int gg_fastrand_query(const float *data, const int *np, int B, int N, const GGGrid &gp, char *wsbase,
                      const GGIndexWs &w, char *scratch, int *nebidx, float *nebmsk, float *cent,
                      float *centmsk, int *centnum, hipStream_t st);
