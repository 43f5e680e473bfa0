// gridgcn_fillgrid.h -- gridgcn_grid_params (include/gridgcn.h) -> GGGrid: the argument checks of the index operators
// (the reference's CHECK_EQs, gridify-inl.h:174-182, plus the limits this implementation relies on).  Host code;
// shared by the C ABI (gridgcn_capi.hip) and by the host-side emulation of the kernels (tests/simt/driver.cpp), so
// that both hand the kernels the same GGGrid.
#pragma once
#include "../../include/gridgcn.h"
#include "gridgcn_dev.h"

static inline int gg_fill_grid(const gridgcn_grid_params *p, int B, int N, bool up, GGGrid *gp)
{
    if (!p || B < 1 || N < 1) return GRIDGCN_EINVAL;
    if (p->max_p_grid < 1 || p->max_p_grid > GG_PMAX) return GRIDGCN_EINVAL;
    if (p->max_o_grid < 1) return GRIDGCN_EINVAL;
    if (p->kernel_size < 1 || p->kernel_size > GG_KMAX || (p->kernel_size & 1) == 0)
        return GRIDGCN_EINVAL;
    long long G = 1;
    for (int j = 0; j < 3; j++) {
        if (p->grid_size[j] < 1 || !(p->voxel_size[j] > 0.0f)) return GRIDGCN_EINVAL;
        G *= p->grid_size[j];
        if (G >= (1ll << 24)) return GRIDGCN_EINVAL;  // the reference indexes voxels in fp32
        gp->shift[j] = p->coord_shift[j];
        gp->vs[j] = p->voxel_size[j];
        gp->rvs[j] = 1.0f / p->voxel_size[j];
        gp->g[j] = p->grid_size[j];
    }
    const long long k3 = (long long)p->kernel_size * p->kernel_size * p->kernel_size;
    if ((long long)B * G >= (1ll << 31) || (long long)B * N >= (1ll << 31)) return GRIDGCN_EINVAL;
    if ((long long)B * p->max_o_grid * p->max_p_grid >= (1ll << 31)) return GRIDGCN_EINVAL;
    if (up && (long long)B * N * k3 >= (1ll << 31)) return GRIDGCN_EINVAL;  // int threadindex
    gp->G = (int)G;
    gp->gxy = p->grid_size[0] * p->grid_size[1];
    gp->P = p->max_p_grid;
    gp->O = p->max_o_grid;
    gp->k = p->kernel_size;
    gp->k3 = (int)k3;
    gp->loc = p->loc;
    gp->seed = p->seed;
    gp->seed_dev = (const unsigned long long *)p->seed_dev;
    return GRIDGCN_OK;
}
