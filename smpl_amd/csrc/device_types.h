// smpl_amd/csrc/device_types.h -- plain-data layout of everything the gfx950
// kernels read: the compiled robot model (flat arrays, the form
// sbpl_collision_checking/src/robot_collision_model.cpp:117-623 reduces a URDF to),
// the voxel grid, the BFS grid, the motion primitives and the goal.
#pragma once

#ifndef __HIPCC_RTC__   // hiprtc (per-robot specialisation, specialize.cpp) brings its own runtime declarations
#include <stdint.h>
#else
typedef signed char int8_t; typedef unsigned char uint8_t; typedef short int16_t; typedef unsigned short uint16_t;
typedef int int32_t; typedef unsigned int uint32_t; typedef long long int64_t; typedef unsigned long long uint64_t;
typedef unsigned long size_t;
#endif

#define SMPLX_MAX_VARS 16
#define SMPLX_MAX_JOINTS 40
#define SMPLX_MAX_NODES 128
// per-thread traversal stack of the sphere-tree walks (LDS), in bytes: what the model's trees need (SmplxModelDev::stack_bytes,
// worked out by the model compiler), at least MIN; a model that needs more than MAX is refused
#define SMPLX_STACK_MIN 16
#define SMPLX_STACK_MAX 64
#define SMPLX_MAX_TREES 24
#define SMPLX_MAX_PAIRS 160
#define SMPLX_MAX_PRIMS 64
#define SMPLX_MAX_SLOTS 2

// joint transform kinds (sbpl_collision_checking/src/transform_functions.h:95-258)
enum { SMPLX_TK_FIXED = 0, SMPLX_TK_REV_X = 1, SMPLX_TK_REV_Y = 2, SMPLX_TK_REV_Z = 3, SMPLX_TK_REV_GENERIC = 4,
       SMPLX_TK_PRISMATIC = 5,
       // the same joints when the origin's rotation is exactly the identity: the products with 0 and 1 are skipped,
       // every non-zero result is bit-identical to the general form
       SMPLX_TK_FIXED_T = 6, SMPLX_TK_REV_X_T = 7, SMPLX_TK_REV_Y_T = 8, SMPLX_TK_REV_Z_T = 9 };
// joint types
enum { SMPLX_JT_FIXED = 0, SMPLX_JT_REVOLUTE = 1, SMPLX_JT_CONTINUOUS = 2, SMPLX_JT_PRISMATIC = 3 };
// where a joint's parent-link transform comes from in the depth-first joint list
enum { SMPLX_SRC_RUNNING = -1, SMPLX_SRC_ROOT = -2 };
// motion primitive types (smpl/include/smpl/graph/motion_primitive.h:48-67)
enum { SMPLX_MP_LONG = -1, SMPLX_MP_SNAP_RPY = 0, SMPLX_MP_SNAP_XYZ = 1, SMPLX_MP_SNAP_XYZ_RPY = 2, SMPLX_MP_SHORT = 3 };
// goal types (smpl/include/smpl/types.h)
enum { SMPLX_GOAL_XYZ = 0, SMPLX_GOAL_XYZ_RPY = 1, SMPLX_GOAL_JOINT = 2 };
// per-successor flags
enum { SMPLX_F_VALID = 1, SMPLX_F_GOAL = 2, SMPLX_F_INACTIVE = 0x10, SMPLX_F_LIMITS = 0x20, SMPLX_F_COLLISION = 0x40,
       SMPLX_F_DEFERRED = 0x80 /* internal: edge did not fit the work list, resolved by a fused pass */ };

struct SmplxJoint {          // depth-first pre-order; the child link of joint i is "link i"
    double origin[12];       // row-major 3x4
    double axis[3];
    int32_t kind;            // SMPLX_TK_*
    int32_t var;             // planning variable feeding the joint, -1 = none (fixed, or held at 0)
    int32_t src;             // SMPLX_SRC_RUNNING / SMPLX_SRC_ROOT / >= 0: restore parent transform from that slot
    int32_t save_slot;       // >= 0: child link transform is kept in that slot for later siblings
    int32_t tree;            // sphere tree on the child link (index into tree_first), -1 = none
    int32_t on_chain;        // 1 if the joint lies between the root and the planning link
};

struct SmplxNode {           // sphere-tree node; trees are stored post-order, root last
    double c[3];
    double r;
    int32_t left, right;     // global node indices, -1/-1 for a leaf
    int32_t thr;             // smallest squared cell distance that clears the sphere (bound to a grid)
    int32_t pad;
};

struct alignas(16) SmplxModelDev {
    int32_t njoints, nvars, ntrees, nnodes, npairs, nslots, nchain, nroot;   // nroot: LDS root-position slots
    int32_t stack_bytes, pad_[3];             // traversal stack per thread (SMPLX_STACK_MIN .. MAX)
    SmplxJoint joints[SMPLX_MAX_JOINTS];
    SmplxNode nodes[SMPLX_MAX_NODES];
    int32_t tree_first[SMPLX_MAX_TREES + 1];
    int32_t tree_joint[SMPLX_MAX_TREES];      // joint whose child link carries the tree
    int32_t pair_a[SMPLX_MAX_PAIRS], pair_b[SMPLX_MAX_PAIRS];  // tree indices, a before b in group order
    // the same pairs regrouped for the kernels: for tree t (in depth-first order of its link) the partners that come
    // EARLIER in that order are pair_other[pair_first[t] .. pair_first[t+1]); a tree that is somebody's earlier
    // partner keeps its root position in LDS slot tree_root_slot[t] (-1 otherwise)
    int32_t tree_root_slot[SMPLX_MAX_TREES];
    int32_t pair_first[SMPLX_MAX_TREES + 1];
    int32_t pair_other[SMPLX_MAX_PAIRS];
    // per planning variable
    double var_min[SMPLX_MAX_VARS], var_max[SMPLX_MAX_VARS];   // planning limits (continuous: -pi, pi)
    double var_min_norm[SMPLX_MAX_VARS];                        // normalize_angle(var_min)
    double var_k[SMPLX_MAX_VARS];                               // motion-sphere factor of the owning joint
    double coord_delta[SMPLX_MAX_VARS];
    int32_t coord_vals[SMPLX_MAX_VARS];
    int32_t var_type[SMPLX_MAX_VARS];                           // SMPLX_JT_*
};

struct SmplxGridDev {
    double origin_minus_res[3];   // origin - res, as distance_map.hpp:525-527 forms it
    double res, inv_res;
    int32_t n[3];                 // interior cells
    int32_t bricks[3];            // 4x4x4 bricks per axis
    int32_t dmax_sqrd, pad;
    const uint16_t* d2;           // brick-tiled squared cell distances
};

// BFS_3D distance grid (bfs3d.h:213-220 holds it as a padded x-fastest array) in BRICK-MAJOR records: an 8x8x8 brick of
// cells is one 4 KB record -- its 512 cells, then copies of its six faces, then of its four z-parallel edges -- so that a
// brick sweep (k_bfs_brick_wave) reads its own cells in one run and every piece of its one-cell halo as a contiguous run of
// a neighbour's record (the x-fastest array gave it 100 rows of 40 bytes: 12x the algorithmic traffic, round 2).
//   [0, 512)     cell (lx, ly, lz) at lz * 64 + ly * 8 + lx
//   [512, 896)   faces x = 0, x = 7 (index lz * 8 + ly), y = 0, y = 7 (lz * 8 + lx), z = 0, z = 7 (ly * 8 + lx), 64 each
//   [896, 928)   edges (x, y) = (0, 0), (7, 0), (0, 7), (7, 7), index lz
//   [928, 936)   the eight cells DIAGONALLY next to the brick's corners (they belong to eight other bricks, which write
//                them here when they change: a visit reads one 32-byte sector of its own record instead of eight of others'),
//                index cx | cy << 1 | cz << 2 (0 = the low side)
// WALL 0x7FFFFFFF, UNDISCOVERED -1 as in the reference (bfs3d.h:48-51); cells beyond the grid in the last bricks are walls.
// A distance carries the TAG of the BFS run that wrote it in its bits 28-30 (tag_word = tag << 28, tag 1..7, tag_mask =
// 0xF0000000): a cell whose tag is not the current run's counts as UNDISCOVERED, so a new goal needs no pass over the records
// to reset them (BFS_3D::run's reset loop, bfs3d.cpp:162-166, was a tenth of a goal at 512^3) except every seventh, when the
// tags wrap.  -1 reads as tag 15, which no run has.  Grids of 2^28 cells or more run with tag_mask = 0 and a reset per goal.
#define SMPLX_BFS_REC 1024
#define SMPLX_BFS_FACES 512
#define SMPLX_BFS_EDGES 896
#define SMPLX_BFS_CORNERS 928
#define SMPLX_BFS_USED 936
struct SmplxBfsDev {
    int32_t dim_x, dim_y, dim_z, dim_xy;     // padded dims (bfs3d.cpp:61-66): what inBounds compares with
    int32_t cost_per_cell, tag_word;
    int32_t nbx, nby, nbz, tag_mask;         // bricks per axis
    const int32_t* dist;                      // nbx * nby * nbz records
};

struct SmplxActionsDev {
    int32_t nprims, use_long_and_short, xy_rotate_by_var3, pad;
    int32_t enabled[4];
    double thresh[4];
    int32_t type[SMPLX_MAX_PRIMS];
    int32_t cost[SMPLX_MAX_PRIMS];            // (int)(1000 * weight)
    double delta[SMPLX_MAX_PRIMS][SMPLX_MAX_VARS];
};

// Device copy of the ManipLattice state table (manip_lattice.cpp:1302-1354: coordinate -> state id), open addressing
// with linear probing.  A slot is `stride` int32: [id + 1 (0 = empty), coord[nvars], padding]; it holds only states the
// host has committed (ids are assigned in the caller's sequential order), so a hit is always right and a miss only
// means "not committed when the table was last synchronised".
struct SmplxTableDev {
    int32_t* slots;
    uint32_t mask;        // capacity - 1 (capacity is a power of two)
    int32_t stride;       // int32 per slot: nvars + 1 rounded up to a multiple of 8 (32-byte sectors)
    int32_t pad;
};

// Compacted successor stream of one frontier batch (K5): every VALID successor leaves an 8-byte record {id, meta} in
// region A -- id = state id if the coordinate is in the device table, else -1; meta = primitive | goal << 8 |
// state index in the batch << 9 -- and, when the host needs the full data (unknown coordinate, or a goal successor
// whose own joint values extractPath reports), a record [h, coord[nvars], pad][q[nvars]] in region B.  A block
// claims one range of each region (wave ballots -> block totals -> one atomic per region), so records appear in
// (state, primitive) order inside a block; block_tab[4b..4b+3] = {first A, count A, first B, count B} of block b.
// Same-address atomics serialise at ~12 ns on this chip (800 blocks on one counter: 19 us measured), so each region is
// cut into SMPLX_CMP_SHARDS sub-regions with their own counters on separate 128-byte lines; block b claims in shard
// b % SMPLX_CMP_SHARDS.  totals[32 k] / totals[32 k + 1] = records of shard k in A / B (its sub-region starts at
// k * cap / SMPLX_CMP_SHARDS); totals[32 * SMPLX_CMP_SHARDS] = 1 if a sub-region overflowed (dense outputs stay valid).
#define SMPLX_CMP_SHARDS 16
#define SMPLX_CMP_TOTALS (32 * SMPLX_CMP_SHARDS + 1)
struct SmplxCompactDev {
    int32_t* rec_a;              // 2 int32 per record
    unsigned char* rec_b;        // rec_b_bytes per record
    int32_t* block_tab;
    int32_t* totals;             // SMPLX_CMP_TOTALS int32, zeroed by the first kernel of the same launch sequence
    int32_t cap_a, cap_b;        // records per region (multiples of SMPLX_CMP_SHARDS)
    int32_t rec_b_bytes, pad;
};

struct SmplxGoalDev {
    int32_t type, pad;
    double angles[SMPLX_MAX_VARS];
    double angle_tol[SMPLX_MAX_VARS];
    int32_t coord[SMPLX_MAX_VARS];
    double xyz[3];
    double xyz_tol[3];
};

static_assert(sizeof(SmplxModelDev) % 16 == 0 && sizeof(SmplxJoint) % 16 == 0 && sizeof(SmplxNode) % 16 == 0,
              "the model is copied to LDS in 16-byte pieces");

// Packed copy of the USED part of the model, the form the kernels stage into LDS: a 64-byte header of int32
// (counts, then byte offsets of the segments) followed by 16-byte aligned segments.
enum { SMPLX_BH_NJOINTS = 0, SMPLX_BH_NVARS, SMPLX_BH_NTREES, SMPLX_BH_NNODES, SMPLX_BH_NPAIRS, SMPLX_BH_NSLOTS,
       SMPLX_BH_NROOT, SMPLX_BH_BYTES, SMPLX_BH_OFF_JOINTS, SMPLX_BH_OFF_NODES, SMPLX_BH_OFF_INTS, SMPLX_BH_OFF_VARD,
       SMPLX_BH_OFF_VARI, SMPLX_BH_STACK, SMPLX_BH_WORDS = 16 };
#define SMPLX_MAX_BLOB_BYTES (64 + SMPLX_MAX_JOINTS * 144 + SMPLX_MAX_NODES * 48 + 4096)

struct SmplxSearchDev;

// everything one query needs, resident in HBM
struct SmplxSpaceDev {
    SmplxModelDev model;
    alignas(16) unsigned char model_blob[SMPLX_MAX_BLOB_BYTES];
    SmplxGridDev grid;
    SmplxBfsDev bfs;
    SmplxActionsDev actions;
    SmplxGoalDev goal;
    SmplxTableDev table;
    SmplxSearchDev* search;       // device-resident ARA* of this query (null until the first device search)
};

// ---------------------------------------------------------------------------------------------------------------------
// Device-resident ARA* (kernels: search_kernel.h, host side: engine.hip).  One persistent workgroup owns one
// query: OPEN (the reference's intrusive binary heap, intrusive_heap.hpp:346-395), INCONS, the state table, the search
// states and the committed successor lists all live in HBM (the top of the heap in LDS while the kernel runs); the host
// only launches, grows buffers when the kernel asks, and reads results.
// ---------------------------------------------------------------------------------------------------------------------

// search state of one lattice state (ARAStar::SearchState, smpl/include/smpl/search/arastar.h:166-180), 32 bytes = one sector
struct SmplxSState {
    uint32_t g, h, f, eg;
    int32_t bp;
    int32_t heap_index;           // position in OPEN, 0 = not there (intrusive_heap.hpp:145-166)
    uint16_t iteration_closed, call_number;
    uint32_t flags;               // bit 0: pushed into OPEN while already in it (INCONS holds a state once per improvement)
};

struct SmplxHeapEntry { uint32_t f; int32_t id; };   // f = the state's f (kept equal to it at all times)
struct SmplxSucc { int32_t id; int32_t cost_prim; };  // cost | primitive << 24 (cost = int(1000 * weight) < 2^24)

// why a launch of k_search came back
enum { SMPLX_SS_RUNNING = 0,      // step budget of the launch used up: launch again
       SMPLX_SS_DONE = 1,         // replan finished (solved or not)
       SMPLX_SS_GROW = 2,         // a buffer is too small for the next expansion: the host enlarges them and launches again
       SMPLX_SS_ERROR = 3 };

struct SmplxSearchDev {
    // ---- buffers (HBM) and their capacities
    int32_t* coord;               // [cap_states][nvars]
    double* q;                    // [cap_states][nvars]      first creator's joint values (manip_lattice.cpp:1329-1354)
    SmplxSState* st;              // [cap_states]
    SmplxHeapEntry* heap;         // [cap_heap + 1], entry 0 unused
    int32_t* incons;              // [cap_incons]
    int32_t* log;                 // [cap_log]                expansion log (state ids in pop order)
    int32_t* done_off;            // [cap_states]             first committed successor, -1 = never expanded
    int32_t* done_cnt;            // [cap_states]             valid successors | evaluated primitives << 8
    SmplxSucc* succ;              // [cap_succ]
    int32_t* path;                // [cap_path]               solution, goal first (bp chain)
    int32_t cap_states, cap_heap, cap_incons, cap_log, cap_succ, cap_path;
    // ---- search parameters (smplx_search_params)
    double initial_eps, final_eps, delta_eps;
    int32_t improve, bounded, max_init, max_rep;
    int32_t start_id, pad0;
    // ---- search state that survives between launches
    double curr_eps, satisfied_eps;
    int32_t nstates, heap_size, n_incons, n_log, n_succ, n_path;
    int32_t iteration, call_number, phase, status;
    int32_t num, expand_count, expand_count_init, err, solved, cost;
    int32_t dup_pushes, grow_what;
    uint32_t goal_f, pad1;
    int64_t committed_evals, gpu_evals, lookups;
    int64_t ticks[8];             // 100 MHz wall clock per phase of thread 0: select, pop, evaluate, commit, relax, reorder, idle
};

// hash of a discretised coordinate (host inserts and device lookups must agree; state ids never depend on it)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t smplx_coord_hash(const int32_t* c, int n)
{
    uint32_t h = 2166136261u;
    for (int i = 0; i < n; ++i) h = (h ^ (uint32_t)c[i]) * 16777619u;
    h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
static inline int smplx_table_stride(int nvars) { return (nvars + 1 + 7) / 8 * 8; }
static inline int smplx_rec_b_bytes(int nvars) { return ((nvars + 1 + 1) / 2 * 2) * 4 + nvars * 8; }
