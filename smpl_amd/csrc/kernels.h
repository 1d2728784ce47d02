// smpl_amd/csrc/kernels.h -- launch geometry and prototypes of the gfx950 kernels (kernels.hip)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.h"

#define SMPLX_BLOCK 128          // 2 waves; per-thread LDS scratch keeps ~4 blocks per CU resident
#define SMPLX_STACK_BYTES 32     // per-thread DFS stack (node indices, one byte each)

// dynamic LDS bytes a collision kernel needs for a model
static inline size_t smplx_lds_bytes(int nnodes, int ntrees, int nslots)
{
    return (size_t)nnodes * sizeof(SmplxNode) + (size_t)(3 * ntrees + 12 * nslots) * 8 * SMPLX_BLOCK +
           (size_t)SMPLX_STACK_BYTES * SMPLX_BLOCK;
}

extern "C" {
__global__ void k_state_prep(const SmplxSpaceDev* S, const double* Q, const int64_t* refs, int B, double* goal_dist,
                             unsigned char* parent_valid, int* parent_lookups);
__global__ void k_expand(const SmplxSpaceDev* S, const double* Q, const int64_t* refs, int B, const double* goal_dist,
                         const unsigned char* parent_valid, const int* parent_lookups, unsigned char* out_flags,
                         int* out_coord, double* out_q, int* out_h, int* out_cost, int* out_lookups,
                         unsigned long long* counters);
__global__ void k_edge_valid(const SmplxSpaceDev* S, const double* Aq, const double* Bq, int n, unsigned char* out,
                             int* out_lookups, int* out_waypoints);
__global__ void k_state_valid(const SmplxSpaceDev* S, const double* Q, int n, unsigned char* out, int* out_lookups);
__global__ void k_heuristic(const SmplxSpaceDev* S, const double* Q, int n, int* out_h, double* out_xyz);
__global__ void k_sphere_positions(const SmplxSpaceDev* S, const double* Q, int n, double* out);
__global__ void k_bfs_init(SmplxGridDev g, int wall_thr, int dim_x, int dim_y, int dim_z, int* dist);
__global__ void k_bfs_reset(int* dist, size_t total);
__global__ void k_bfs_seed(int* dist, int origin, int* queue, int* counts);
__global__ void k_bfs_level(int* dist, const int* q_in, int* q_out, int* counts, int level, int dim_x, int dim_xy);
}
