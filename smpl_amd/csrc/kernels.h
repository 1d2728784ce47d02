// smpl_amd/csrc/kernels.h -- launch geometry and prototypes of the gfx950 kernels (kernels.hip)
#pragma once

#ifndef __HIPCC_RTC__   // hiprtc (per-robot specialisation, specialize.cpp) brings its own runtime declarations
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "device_types.h"

#define SMPLX_BLOCK 128          // 2 waves; per-thread LDS scratch keeps ~4 blocks per CU resident
#define SMPLX_SEARCH_STATIC_LDS (44 * 1024)   // static LDS of k_search (2 x ExpandLds + SearchLds + header and primitives copies), an upper bound
#define SMPLX_TALLIES 6           // per-block tallies (tally_block)     // per-thread DFS stack (node indices, one byte each)

// dynamic LDS bytes: the packed model, plus (collision kernels) per-thread scratch
// small-batch kernel: 7 waypoint lanes per primitive + the state's own lane in whole "config" waves, then one more wave
// whose lanes are the primitives' bookkeeping lanes + the goal-distance lane (nprims + 1 <= 64)
static inline int smplx_small_block(int nprims) { return ((nprims * 7 + 1) + 63) / 64 * 64 + 64; }
// k_search: the same, plus a wave that works out the successors' heuristics beside the search wave where 512 threads allow it
static inline int smplx_search_block(int nprims) { const int b = smplx_small_block(nprims); return b + 64 <= 512 ? b + 64 : b; }
static inline size_t smplx_lds_bytes_n(size_t blob_bytes, int nroot, int nslots, int nvars, int stack_bytes, int nthreads)
{
    return blob_bytes + (size_t)(3 * nroot + 12 * nslots + nvars) * 8 * nthreads + (size_t)stack_bytes * nthreads;
}
static inline size_t smplx_lds_bytes(size_t blob_bytes, int nroot, int nslots, int nvars, int stack_bytes)
{
    return smplx_lds_bytes_n(blob_bytes, nroot, nslots, nvars, stack_bytes, SMPLX_BLOCK);
}

extern "C" {
__global__ void k_state_prep(const SmplxSpaceDev* S, const double* Q, const int64_t* refs, int B, double* goal_dist,
                             unsigned char* parent_valid, int* parent_lookups,
                         const SmplxSpaceDev* const* stab, const unsigned short* state_q);
__global__ void k_expand(const SmplxSpaceDev* S, const double* Q, const int64_t* refs, int B, const double* goal_dist,
                         const unsigned char* parent_valid, const int* parent_lookups, unsigned char* out_flags,
                         int* out_coord, double* out_q, int* out_h, int* out_cost, int* out_lookups,
                         unsigned long long* counters, const int* deferred_count,
                         const SmplxSpaceDev* const* stab, const unsigned short* state_q);
__global__ void k_pipe_prep(const SmplxSpaceDev* S, const double* Q, const int64_t* refs, int B, double* goal_dist,
                            int* work_count,
                         const SmplxSpaceDev* const* stab, const unsigned short* state_q, int* cmp_totals,
                            const int* ins_items, int n_ins);
__global__ void k_pipe_setup(const SmplxSpaceDev* S, const double* Q, const int64_t* refs, int B, const double* goal_dist,
                             unsigned char* out_flags, double* out_q, int* edge_w, int* edge_lookups,
                             unsigned char* edge_bad, int* state_lookups, unsigned char* state_bad, unsigned long long* work,
                             int* work_count, int capacity,
                         const SmplxSpaceDev* const* stab, const unsigned short* state_q);
__global__ void k_pipe_configs(const SmplxSpaceDev* S, const double* Q, const int64_t* refs, int B, const double* out_q,
                               const int* edge_w, int* edge_lookups, unsigned char* edge_bad, int* state_lookups,
                               unsigned char* state_bad, const unsigned long long* work, const int* work_count, int capacity);
__global__ void k_pipe_finish(const SmplxSpaceDev* S, const double* Q, const int64_t* refs, int B, const int* edge_w,
                              const int* edge_lookups, const unsigned char* edge_bad, const int* state_lookups,
                              const unsigned char* state_bad, unsigned char* out_flags, int* out_coord, double* out_q,
                              int* out_h, int* out_cost, int* out_lookups, unsigned long long* counters,
                              const double* goal_dist,
                         const SmplxSpaceDev* const* stab, const unsigned short* state_q, int* out_id, SmplxCompactDev cmp);
__global__ void k_small_batch(const SmplxSpaceDev* S, const double* Q, const int64_t* refs, int B, double* goal_dist_out,
                              unsigned char* state_bad_out, int* state_lookups_out, unsigned char* out_flags, int* out_coord,
                              double* out_q, int* out_h, int* out_cost, int* out_lookups,
                              const SmplxSpaceDev* const* stab, const unsigned short* state_q, unsigned char* host_flags,
                              int* host_coord, double* host_q, int* host_h, int* out_id, int* host_id, const int* ins_items,
                              int n_ins);
__global__ void k_search(const SmplxSpaceDev* const* stab, int max_steps, int lh, int* status_out);
__global__ void k_search_table_fill(const SmplxSpaceDev* Sq, const int* coord, int first, int n, int nvars);
__global__ void k_heap_ops(const int* ops, int nops, int lh, unsigned long long* heap_hbm, SmplxSState* st, int* top_after);
__global__ void k_edge_valid(const SmplxSpaceDev* S, const double* Aq, const double* Bq, int n, unsigned char* out,
                             int* out_lookups, int* out_waypoints);
__global__ void k_state_valid(const SmplxSpaceDev* S, const double* Q, int n, unsigned char* out, int* out_lookups);
__global__ void k_heuristic(const SmplxSpaceDev* S, const double* Q, int n, int* out_h, double* out_xyz);
__global__ void k_sphere_positions(const SmplxSpaceDev* S, const double* Q, int n, double* out);
__global__ void k_table_insert(const SmplxSpaceDev* S, const SmplxSpaceDev* const* stab, const int* items, int n, int nvars);
__global__ void k_bfs_metric(SmplxGridDev grid, SmplxBfsDev bfs, const double* xyz, int n, double* out);
__global__ void k_bfs_init(SmplxGridDev g, int wall_thr, int nbx, int nby, int nbz, int* dist);
__global__ void k_bfs_reset(int* dist, size_t total);
__global__ void k_bfs_export(SmplxBfsDev b, int* out);
__global__ void k_bfs_brick_seed(int* dist, int cx, int cy, int cz, int nbx, int nby, int nbz, int* list0, int* counts, int tag_word);
__global__ void k_bfs_brick_wave(int* dist, int nbx, int nby, int nbz, const int* list_in, const int* counts_in, int* list_next,
                                 int* counts_next, int* counts_after, int shard_cap, int* queued_mine, int* queued_next, int* queue_size_out, int tag_word, int tag_mask);
}
