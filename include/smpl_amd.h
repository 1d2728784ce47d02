/* include/smpl_amd.h -- C-ABI of the MI355X-native ARA* state-expansion engine for smpl.
 *
 * This is the drop-in boundary: plain pointers and sizes, opaque handles, int status codes,
 * nothing thrown across it.  Each entry point names the reference interface it stands behind
 * (paths relative to the dyouakim/smpl tree).  INTEGRATION.md shows the C++ shims a maintainer
 * adds on the smpl side (include/smpl_amd/plugin.hpp holds them ready-made).
 *
 * Conventions
 *   - every function returns SMPLX_OK (0) or a negative SMPLX_E_* code; smplx_last_error() gives text
 *   - q / state arrays hold the planning variables in planning-joint order (RobotState,
 *     smpl/include/smpl/types.h:66)
 *   - host pointers are copied during the call; "_device" variants take pointers into HBM and a
 *     hipStream_t (passed as void*) and do not synchronise
 *   - joint values handed in through host pointers must be finite with |q| < 1e6 (SMPLX_E_ARG otherwise: the limit
 *     folding of KDLRobotModel::checkJointLimits would not terminate on them); "_device" variants cannot check
 *   - calls on one handle are serialised by the caller, like the reference's single-threaded
 *     plugins (sbpl_collision_checking/src/collision_space.cpp:741-774 mutates state per query)
 */
#ifndef SMPL_AMD_H
#define SMPL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    SMPLX_OK = 0,
    SMPLX_E_ARG = -1,        /* bad argument */
    SMPLX_E_PARSE = -2,      /* robot / primitive text malformed */
    SMPLX_E_LIMIT = -3,      /* exceeds a compile-time capacity */
    SMPLX_E_HIP = -4,        /* HIP runtime error (no GPU, out of memory, ...) */
    SMPLX_E_STATE = -5,      /* call sequence error (goal/start not set, unknown id) */
    SMPLX_E_INVALID = -6     /* start state violates limits or is in collision */
};

typedef struct smplx_grid smplx_grid;     /* OccupancyGrid, lookup side */
typedef struct smplx_model smplx_model;   /* RobotCollisionModel + RobotModel as flat arrays */
typedef struct smplx_space smplx_space;   /* ManipLattice + BfsHeuristic + CollisionSpace, one query */

const char* smplx_last_error(void);
int smplx_device_count(void);

/* ---- voxel grid ---------------------------------------------------------------------------
 * sbpl::OccupancyGrid / DistanceMap, lookup side
 * (smpl/include/smpl/occupancy_grid.h:56-304; smpl/include/smpl/distance_map/detail/distance_map.hpp:281-300,520-536).
 * d2[nx*ny*nz]: squared cell distance to the nearest occupied or border cell, capped at
 * ceil(max_dist/res)^2 -- the `dist` field of the reference's cells (distance_map.h:115), x-major,
 * z fastest.  Stored in HBM as 16-bit values in 4x4x4 bricks. */
int smplx_grid_create(const double origin[3], int nx, int ny, int nz, double res, double max_dist,
                      const int32_t* d2, smplx_grid** out);
void smplx_grid_destroy(smplx_grid* g);

/* ---- voxel grid, construction side -----------------------------------------------------------------
 * OccupancyGrid::addPointsToField / removePointsFromField (smpl/src/occupancy_grid.cpp:357-422) over
 * DistanceMap::addPointsToMap / removePointsFromMap (distance_map.hpp:306-435): the field is built and kept on the
 * GPU.  Specified as the EXACT Euclidean transform with the reference's cap (ceil(max_dist/res) cells) and border rule
 * (the one-cell layer around the grid is occupied): d2 = min(dmax^2, min over occupied/border cells of |c - o|^2).
 * The reference's own bucketed propagation (distance_map.hpp:627-839) depends on its pop order where distances tie and
 * cannot be compiled here (Eigen): parity with it is unpinned; exactness is tested against brute force.
 * A new field takes three separable passes over the grid; an edit re-runs them on the window its cells can reach. */
int smplx_grid_create_empty(const double origin[3], int nx, int ny, int nz, double res, double max_dist, smplx_grid** out);
/* OccupancyGrid's ref_counted mode (smpl/include/smpl/occupancy_grid.h:66-74, initRefCounts occupancy_grid.cpp:424-441):
 * every cell keeps the number of times it was added; a cell becomes an obstacle when its count leaves 0 and free again
 * when it returns to 0 (a point listed twice counts twice; removing from a cell with count 0 does nothing).  Turning it
 * on gives every occupied cell the count 1. */
int smplx_grid_set_ref_counted(smplx_grid* g, int on);
/* boxes: n x {cx, cy, cz, sx, sy, sz} (world metres), the objects of a scene file (smpl_test/src/call_planner.cpp:158-207):
 * a box occupies the cells from the cell of its low corner to the cell of its high corner */
int smplx_grid_add_boxes(smplx_grid* g, const double* boxes, int n);
/* xyz: n world points (voxel centres); points outside the grid are skipped (distance_map.hpp:312-316) */
int smplx_grid_add_points(smplx_grid* g, const double* xyz, int n);
int smplx_grid_remove_points(smplx_grid* g, const double* xyz, int n);
/* OccupancyGrid::updatePointsInField -> DistanceMap::updatePointsInMap (occupancy_grid.cpp:408-415, distance_map.hpp:367-435):
 * as sets of cells, the obstacles in old \ new are removed and those in new \ old added; like the reference's it does not
 * touch the reference counts */
int smplx_grid_update_points(smplx_grid* g, const double* old_xyz, int n_old, const double* new_xyz, int n_new);
/* the reference counts, x-major / z fastest like d2 (SMPLX_E_STATE when the grid keeps none) */
int smplx_grid_copy_counts(const smplx_grid* g, int32_t* counts);
/* cells the last edit recomputed: the bounding box of the changed cells grown by ceil(max_dist / res) along every axis (an
 * edit is incremental, like the reference's propagation), or the whole grid when that box covers more than half of it */
long long smplx_grid_last_edit_cells(const smplx_grid* g);
/* EDITS AND SPACES.  CollisionChecker queries (smplx_cc_*, smplx_expand_batch) read the field as it is.  A planning space,
 * however, caches successor lists (the reference's ManipLattice re-evaluates every GetSuccs), and its BFS walls are those of
 * the field it was created on (BfsHeuristic::syncGridAndBfs runs once, at init: bfs_heuristic.cpp:52-71, 331-353 -- the
 * reference's heuristic has the same staleness).  After an edit, smplx_get_succs / smplx_plan* on a space of that grid
 * return SMPLX_E_STATE until the goal is set again (which restarts the lattice); re-create the space to get new walls.
 * Coordinates must be finite with |x| < 1e6 (SMPLX_E_ARG). */
/* the field as smplx_grid_create takes it: d2[nx*ny*nz], x-major, z fastest */
int smplx_grid_copy_d2(const smplx_grid* g, int32_t* d2);

/* ---- robot model --------------------------------------------------------------------------
 * RobotCollisionModel (sbpl_collision_checking/src/robot_collision_model.cpp:117-623),
 * sphere trees (base_collision_models.cpp:337-444), RobotMotionCollisionModel
 * (robot_motion_collision_model.cpp:41-275) and the planning RobotModel limits/FK
 * (sbpl_kdl_robot_model/src/kdl_robot_model.cpp:210-235,400-423) from a plain-text description
 * (format: DESIGN.md section 4). */
int smplx_model_create(const char* robot_text, smplx_model** out);
void smplx_model_destroy(smplx_model* m);
int smplx_model_counts(const smplx_model* m, int* njoints, int* nvars, int* ntrees, int* nnodes, int* npairs, int* nslots);
/* compiled arrays, for inspection: per joint (depth-first order) origin[12], k, file index;
 * per node xyz+r, left, right (global indices); tree_first[ntrees+1]; pairs[2*npairs] */
int smplx_model_joints(const smplx_model* m, double* origins, double* k, int* file_index);
int smplx_model_nodes(const smplx_model* m, double* xyzr, int* left, int* right, int* tree_first);
int smplx_model_pairs(const smplx_model* m, int* pairs);

/* ---- planning space ------------------------------------------------------------------------ */
typedef struct smplx_params {
    double resolutions[16];       /* PlanningParams "discretization" (smpl_ros/src/ros/planner_interface.cpp:149-181) */
    double bfs_inflation_radius;  /* planning_link_sphere_radius (smpl/include/smpl/planning_params.h:71) */
    int32_t cost_per_cell;        /* planning_params.h:68 */
    int32_t use_short_dist_mprims;
    double short_dist_mprims_thresh;
    int32_t use_xyzrpy_snap_mprim;    /* joint-space goals only: the action is the goal itself
                                         (manip_lattice_action_space.cpp:551-559) */
    double xyzrpy_snap_dist_thresh;
    int32_t xy_rotate_by_var3;        /* [FORK] manip_lattice_action_space.cpp:590-599 rotates delta[0], delta[1] by
                                         state[3] in every applyMotionPrimitive: 1 = the reference's behaviour (what
                                         every caller of this repo passes unless it asks for upstream smpl), 0 = upstream */
    int32_t use_long_and_short;
    double padding;                   /* SelfCollisionModel m_padding, default 0 */
    int32_t batch_states;             /* frontier batch B (0 = default 4096) */
    int32_t flags;                    /* SMPLX_SPACE_* bits below; 0 = defaults */
} smplx_params;

/* smplx_params.flags: which kernels serve the space.  Results are identical whatever the bits (the parity suite runs
 * every combination); they exist for A/B measurements and for callers that want the reference's own lookup tallies. */
enum {
    SMPLX_SPACE_FUSED = 1,            /* one GPU thread walks a whole edge in the reference's waypoint order (exact
                                         reference lookup tallies also on colliding edges); default: waypoint-parallel */
    SMPLX_SPACE_NO_SMALL_KERNEL = 4,  /* small batches (<= 512 states) go through the four-kernel pipeline too */
    SMPLX_SPACE_GENERIC_KERNELS = 8   /* no per-robot kernel build (see smplx_space_specialized) */
};

/* RobotPlanningSpace::init + insertHeuristic (smpl/include/smpl/graph/robot_planning_space.h:68,89;
 * smpl/src/graph/manip_lattice.cpp:72-149; smpl/src/heuristic/bfs_heuristic.cpp:52-71,331-353;
 * sbpl_collision_checking/src/collision_space.cpp:689-739).  mprim_text: .mprim file contents
 * (manip_lattice_action_space.cpp:103-195, upstream or fork rows). */
int smplx_space_create(const smplx_model* model, const smplx_grid* grid, const char* mprim_text,
                       const smplx_params* params, smplx_space** out);
void smplx_space_destroy(smplx_space* s);
/* 1 if the space runs kernels compiled for this robot: at creation the model's kinematic structure is turned into
 * compile-time constants and the kernels are built against them with hiprtc (cached per process and on disk under
 * $SMPLX_CACHE_DIR / $XDG_CACHE_HOME/smpl_amd / $HOME/.cache/smpl_amd); results are bit-identical to the generic
 * kernels'.  0: generic kernels (note says why).  Env SMPLX_SPECIALIZE=0 disables, =2 makes a failed build an error. */
int smplx_space_specialized(const smplx_space* s, char* note, int cap);
/* the constants header the per-robot build is compiled against (returns the size needed, including the NUL) */
int smplx_model_const_header(const smplx_model* m, char* out, int cap);
int smplx_space_num_vars(const smplx_space* s);
int smplx_space_num_prims(const smplx_space* s);
int smplx_space_discretization(const smplx_space* s, int32_t* coord_vals, double* coord_deltas);

/* RobotModel::checkJointLimits as KDLRobotModel implements it (sbpl_kdl_robot_model/src/kdl_robot_model.cpp:173-189,
 * 210-235: fold by 2*pi into [min, min + 2*pi), then compare with the limits); host arithmetic, n states */
int smplx_check_joint_limits(const smplx_space* s, const double* q, int n, uint8_t* ok);

/* ---- CollisionChecker (smpl/include/smpl/collision_checker.h:48-130) ---- */
/* isStateValid (collision_space.cpp:532-536) */
int smplx_cc_state_valid_batch(smplx_space* s, const double* q, int n, uint8_t* valid, int32_t* lookups);
/* same with everything resident in HBM: launches on `stream` (a hipStream_t) and returns without synchronising;
 * d_lookups may be NULL.  One launch = n configurations through FK + sphere trees vs grid + checked link pairs: the
 * "K2" collision micro-benchmark of sbpl_collision_checking_test/src/benchmark_cc.cpp:234-256 is a loop over this. */
int smplx_cc_state_valid_batch_device(smplx_space* s, const double* d_q, int n, uint8_t* d_valid, int32_t* d_lookups,
                                      void* stream);
/* isStateToStateValid (collision_space.cpp:538-581); lookups/waypoints may be NULL */
int smplx_cc_edge_valid_batch(smplx_space* s, const double* a, const double* b, int n, uint8_t* valid,
                              int32_t* lookups, int32_t* waypoints);
/* interpolatePath (collision_space.cpp:583-640): waypoints of the edge a->b, out holds cap*nvars doubles */
int smplx_cc_interpolate(smplx_space* s, const double* a, const double* b, double* out, int cap, int* n);
/* world positions of all sphere-tree nodes for n states: out[n][nnodes][3] (RobotCollisionState::updateSphereState) */
int smplx_cc_sphere_positions(smplx_space* s, const double* q, int n, double* out);

/* ---- RobotHeuristic / BfsHeuristic (smpl/include/smpl/heuristic/robot_heuristic.h:53-101) ---- */
/* RobotPlanningSpace::setGoal with a JOINT_STATE_GOAL (manip_lattice.cpp:2248-2287; goal pose = FK of the
 * angles, planner_interface.cpp:1232-1235) -> BfsHeuristic::updateGoal -> BFS_3D::run to completion */
int smplx_set_goal_joint(smplx_space* s, const double* angles, const double* tolerances);
/* XYZ_GOAL (manip_lattice.cpp:1672-1687) */
int smplx_set_goal_xyz(smplx_space* s, const double xyz[3], const double tol[3]);
int smplx_goal_pose(const smplx_space* s, double xyz[3]);
/* GetGoalHeuristic for arbitrary states (bfs_heuristic.cpp:148-163); xyz (n*3) may be NULL */
int smplx_heuristic_batch(smplx_space* s, const double* q, int n, int32_t* h, double* xyz);
/* BfsHeuristic::getMetricGoalDistance (bfs_heuristic.cpp:129-138) for n workspace points xyz[n*3]:
 * BFS cell distance * resolution, WALL * resolution outside the grid.  This is the value the action space gates the
 * short / snap primitives on (manip_lattice_action_space.cpp:393-397). */
int smplx_bfs_metric_goal_distance(smplx_space* s, const double* xyz, int n, double* out);
/* BfsHeuristic::getMetricStartDistance (bfs_heuristic.cpp:103-127): Manhattan cell distance to the cell of the start
 * state's planning link, times the resolution (needs smplx_set_start) */
int smplx_bfs_metric_start_distance(smplx_space* s, const double* xyz, int n, double* out);
/* the padded (nx+2)(ny+2)(nz+2) BFS_3D distance grid, node order of bfs3d.h:213-220 */
int64_t smplx_bfs_size(const smplx_space* s);
int smplx_bfs_copy(smplx_space* s, int32_t* out);
/* passes the last BFS took: sweeps over the queued 8x8x8 bricks (the device keeps the grid in brick-major records) */
int smplx_bfs_levels(const smplx_space* s);

/* ---- ManipLattice (smpl/include/smpl/graph/manip_lattice.h:63-307) ---- */
/* successor evaluation for B arbitrary parent states: the loop body of GetSuccs
 * (manip_lattice.cpp:254-305) for every (state, primitive), dense outputs indexed [state][prim]:
 * flags (SMPLX_F_* of device_types.h: 1 valid, 2 goal, 0x10 inactive, 0x20 limits, 0x40 collision),
 * coord[nvars], q[nvars], h, cost, lookups.  Any output may be NULL.
 * Defined entries: flags everywhere; coord, h and cost where flags has the valid bit (h and cost are 0 elsewhere); q on every
 * edge that is not inactive; lookups on every edge not in collision (a colliding edge's tally stops at its first colliding
 * waypoint lane).  Other entries hold whatever the kernel that served the batch left there. */
int smplx_expand_batch(smplx_space* s, const double* q, int B, uint8_t* flags, int32_t* coord, double* succ_q,
                       int32_t* h, int32_t* cost, int32_t* lookups);
/* same, everything resident in HBM; launches on `stream` and returns without synchronising.
 * work: device scratch of smplx_expand_work_bytes(s, B) bytes.  d_counters: see smplx_counters_bytes (may be NULL). */
size_t smplx_expand_work_bytes(const smplx_space* s, int B);
/* d_counters: device scratch of smplx_counters_bytes(s, B) bytes, zeroed by the caller; tallies accumulate per
 * thread block without atomics.  smplx_counters_read sums them (synchronous copy): out[0] successor evaluations,
 * out[1] valid successors, out[2] grid lookups the reference algorithm makes for those edges, out[3] grid lookups
 * the edge kernels themselves issued (waypoint 0 is checked once per state), out[4] configurations checked by
 * k_pipe_configs (states + waypoints; 0 in fused mode), out[5] lookups of the per-state checks among them. */
size_t smplx_counters_bytes(const smplx_space* s, int B);
int smplx_counters_read(const smplx_space* s, const uint64_t* d_counters, int B, uint64_t out[6]);
int smplx_expand_batch_device(smplx_space* s, const double* d_q, int B, uint8_t* d_flags, int32_t* d_coord,
                              double* d_succ_q, int32_t* d_h, int32_t* d_cost, int32_t* d_lookups,
                              void* d_work, uint64_t* d_counters, void* stream);

/* ---- K5: state table on the device + compacted successor stream --------------------------------------------------
 * ManipLattice::getHashEntry / createHashEntry (manip_lattice.cpp:1302-1354).  Every space keeps a device copy of its
 * coordinate -> id table (open addressing over 32-byte slots).  Ids are still ASSIGNED by the host in the caller's
 * sequential commit order (that order is what makes them the reference's ids); the states committed since the last
 * batch are inserted on the device at the head of the next one.  The successor kernels look every valid successor's
 * coordinate up: a hit hands the id back with the batch and the host skips its own lookup when it commits the
 * successor; a miss (-1) means "not committed when the batch ran" and the host does getOrCreateState as before.
 * With the compact arguments the same launch also leaves the VALID successors as a stream built with wavefront
 * ballots (one atomic per region and thread block):
 *   region A, 8 bytes per valid successor: {id or -1, primitive | goal << 8 | state index in the batch << 9}
 *   region B, smplx_compact_rec_b_bytes() per successor the host needs in full (unknown coordinate, or goal successor):
 *             int32 h, int32 coord[nvars], padding to 8 bytes, double q[nvars]
 *   block_tab[4 b .. 4 b + 3] = {first A, count A, first B, count B} of thread block b (smplx_compact_blocks(B) blocks);
 *             walking the blocks in order gives the records in (state, primitive) order
 *   Each region is cut into 16 sub-regions with their own counters (same-address atomics serialise on this chip): on
 *   the device d_totals has smplx_compact_totals_len() int32 -- [32 k], [32 k + 1] = records of sub-region k in A, B
 *   (sub-region k starts at k * cap / 16), the last one = 1 if a sub-region was too small (the dense outputs are still
 *   complete); the host-pointer form returns totals[0], totals[1] = records in A, B and totals[2] = that flag
 * smplx_table_sync creates the device table if the space has none yet and pushes pending inserts without a batch; so do
 * the two entry points below.  The searches driven by smplx_plan / smplx_plan_multi / smplx_get_succs use the device
 * table only with env SMPLX_DEVICE_TABLE=1 (same results either way): at the batch sizes a search produces the ids it
 * hands back save the host less time than the lookups add to every batch (A/B figures in DESIGN.md section 6). */
int smplx_table_sync(smplx_space* s);
size_t smplx_compact_rec_b_bytes(const smplx_space* s);
int smplx_compact_blocks(const smplx_space* s, int B);
int smplx_compact_totals_len(void);
int smplx_compact_capacity(const smplx_space* s, int B);   /* records per region with which no sub-region can overflow */
/* everything resident in HBM (d_succ_id, and the four compact arguments together, may be NULL); launches on `stream` */
int smplx_expand_batch_k5_device(smplx_space* s, const double* d_q, int B, uint8_t* d_flags, int32_t* d_coord, double* d_succ_q,
                                 int32_t* d_h, int32_t* d_cost, int32_t* d_lookups, int32_t* d_succ_id, int32_t* d_rec_a, int cap_a,
                                 void* d_rec_b, int cap_b, int32_t* d_block_tab, int32_t* d_totals, void* d_work,
                                 uint64_t* d_counters, void* stream);
/* host-pointer form (synchronous): dense flags / coord / succ_q / h / succ_id may be NULL */
int smplx_expand_batch_k5(smplx_space* s, const double* q, int B, uint8_t* flags, int32_t* coord, double* succ_q, int32_t* h,
                          int32_t* succ_id, int32_t* rec_a, int cap_a, void* rec_b, int cap_b, int32_t* block_tab, int32_t totals[3]);

/* per-kernel timing of the next max_launches expand launches with HIP events recorded on the launch
 * stream (no synchronisation until _end): summed milliseconds of k_state_prep and k_expand.
 * Mirrors the ARAStar/ManipLattice stopwatch hooks (smpl/src/profiling.h:56-117). */
int smplx_profile_begin(smplx_space* s, int max_launches);
int smplx_profile_end(smplx_space* s, double* prep_ms, double* expand_ms, int* launches);

/* ManipLattice::setStart (manip_lattice.cpp:1944-1981): limits + collision check, id assigned */
int smplx_set_start(smplx_space* s, const double* q, int* id);
int smplx_start_id(const smplx_space* s);
int smplx_goal_id(const smplx_space* s);   /* always 0 (manip_lattice.cpp:122) */
/* GetSuccs (manip_lattice.cpp:219-313): valid successors of a state id in primitive order */
int smplx_get_succs(smplx_space* s, int id, int32_t* succs, int32_t* costs, int cap, int* n);
/* optional: ids the caller expects to expand soon (top of OPEN); they ride along in the next batch.  A caller that
 * never hints (an unchanged SBPL planner) still gets frontier batches: the space mirrors the g-values the caller's
 * GetSuccs sequence implies (arastar.cpp:546-551) and lets the unevaluated states with the smallest g + w*h ride
 * along on every miss (env SMPLX_AUTO_SPECULATE = states per miss, default 96, 0 = off; w = 5).  Speculation never
 * changes results: ids are assigned when the caller's own sequence commits a state. */
int smplx_hint_frontier(smplx_space* s, const int32_t* ids, int n);
/* GetSuccs has no way to report a failure to an SBPL caller (the successor list just stays empty, which reads as a
 * dead end): the first such error is kept on the space.  Returns SMPLX_OK or that error code; msg gets its text. */
int smplx_space_status(const smplx_space* s, char* msg, int cap);
void smplx_space_clear_status(smplx_space* s);
/* RobotHeuristic::GetGoalHeuristic(state_id) */
int smplx_get_goal_heuristic(smplx_space* s, int id, int32_t* h);
int smplx_num_states(const smplx_space* s);
/* running totals of the space since its goal was last set: out[0] GPU frontier batches, [1] GetSuccs calls served from
 * the cache, [2] GetSuccs calls that had to wait for a batch, [3] committed successor evaluations, [4] successor
 * evaluations the GPU performed (incl. speculative), [5] states ("expands"/"stats" of planner_interface.cpp:1438-1446) */
int smplx_space_counters(const smplx_space* s, int64_t out[6]);
int smplx_get_state(const smplx_space* s, int id, double* q, int32_t* coord);

/* ---- the caller: ARA* (smpl/src/search/arastar.cpp:107-215,486-582) ---- */
typedef struct smplx_search_params {
    double initial_eps, final_eps, delta_eps;
    int32_t improve;              /* ARAStar::setImproveSolution */
    int32_t bounded;              /* expansion bound (TimeParameters::EXPANSIONS) */
    int32_t max_expansions_init, max_expansions;
} smplx_search_params;

typedef struct smplx_search_stats {
    int32_t solved;               /* 1 = success, as ARAStar::replan's return */
    int32_t path_len, cost, expansions, expansions_init;
    double satisfied_eps;
    double seconds;               /* wall time inside replan */
    int64_t gpu_succ_evals;       /* successor evaluations the GPU performed (incl. speculative) */
    int64_t committed_succ_evals; /* evaluations of the committed expansions, counted as the reference performs them:
                                     once per GetSuccs call, so a state re-expanded in a later ARA* iteration counts
                                     again although the engine serves the repeat from its committed list */
    int64_t gpu_batches;
    int64_t cache_hits, cache_misses;
    int64_t grid_lookups;
} smplx_search_stats;

/* ARAStar::replan from scratch (arastar.cpp:107-215) on the query of `s`.
 * Where the search runs.  Two or more queries (smplx_plan_multi), or env SMPLX_SEARCH=device: ON THE DEVICE (SURVEY row
 * N2): one persistent workgroup owns a query -- OPEN (the
 * reference's binary heap with its sift rules, intrusive_heap.hpp:346-395), INCONS, the state table and the search
 * states live in HBM, the workgroup pops a state, evaluates its successors on its lanes, creates states with ids in its
 * own commit order (= the reference's ids) and pushes them; the host only launches, enlarges buffers when asked, and
 * reads results (stats: gpu_batches = kernel launches, cache_misses = 0).  The lattice stays in HBM until an entry point
 * needs it on the host (smplx_get_state, smplx_expansion_log, smplx_get_succs ...), which then see every state the
 * device created.  A single query (it is a chain of dependent expansions: the host loop's speculative batches are the
 * faster way to run one), robots whose expansion does not fit one workgroup's LDS, spaces created with SMPLX_SPACE_FUSED /
 * SMPLX_SPACE_NO_SMALL_KERNEL, and env SMPLX_SEARCH=host take the host-driven loop instead: the same sequential ARA*
 * on the host with frontier batches on the GPU (what an external SBPL planner gets through smplx_get_succs).  Results
 * are identical either way. */
int smplx_plan(smplx_space* s, const smplx_search_params* p, int32_t* path_ids, int cap, smplx_search_stats* stats);
/* diagnostics of the device-resident search of this space: out[0] searches run on the device, [1] buffer enlargements,
 * [2] pushes of a state already in OPEN in the last search (the reference's INCONS holds a state once per improvement),
 * [3..9] 10-ns ticks the search wave of the workgroup spent, last search: (idle), select + pop, until the waypoint lanes'
 * round has closed (ancestor prefetch, preparing the next round, waiting), getOrCreateState, relaxation + pushes, epsilon
 * steps (reorder), load / store of the launch state; [10] states on the device, [11] heap entries cached in LDS;
 * [12] evaluation rounds opened on a guess of the next pop, [13] guesses the pop confirmed; [14..15] 0 */
int smplx_search_counters(const smplx_space* s, int64_t out[16]);
/* nq independent queries (each its own smplx_space: goal, BFS grid, state table) on one GPU.  Device-resident search
 * (default): one workgroup per query, all in ONE launch when the queries share grid, robot and primitives.  Host-driven
 * loop (SMPLX_SEARCH=host): a query runs until it misses, its frontier batch is issued, and the thread moves on to
 * the next query.  No data is exchanged between queries; every query's result equals what smplx_plan gives alone.
 * path_ids holds nq rows of cap ids (may be NULL); stats holds nq entries (seconds = completion time since the
 * start of the call); wall_seconds = duration of the whole call.  Queries created on the same smplx_grid and
 * smplx_model with the same primitives share launches (one cross-query frontier batch per sweep); host_threads > 1
 * splits them into that many slices, each driven by its own host thread.  (BASELINE config 4: 128 queries per GPU.) */
int smplx_plan_multi(smplx_space** spaces, int nq, const smplx_search_params* p, int32_t* path_ids, int cap,
                     smplx_search_stats* stats, double* wall_seconds, int host_threads);

/* Query sharding over ranks, one process per GPU (independent queries partition over GPUs with no data-path
 * collective; the demo's outer loop over requests, smpl_test/src/call_planner.cpp): rank r of `world` owns queries
 * [*first, *first + *count) of a list of `total`, `per_rank` each (BASELINE config 4: 128), the last ranks possibly fewer or
 * none.  A C or C++ caller plans its range with smplx_plan_multi on its own GPU and gathers the per-query results with
 * whatever its job uses (MPI, RCCL: a few integers per query). */
int smplx_shard_range(int rank, int world, int total, int per_rank, int* first, int* count);

int smplx_expansion_log_size(const smplx_space* s);
int smplx_expansion_log(const smplx_space* s, int32_t* out);
/* ManipLattice::extractPath for a plain id path (manip_lattice.cpp:2018-2155): q[len][nvars] */
int smplx_extract_path(smplx_space* s, const int32_t* ids, int len, double* q);

/* --- path post-processing: PlannerInterface::postProcessPath (smpl_ros/src/ros/planner_interface.cpp:2651-2697) over
 * ShortcutPath(JOINT_SPACE) (smpl/src/post_processing.cpp:281-309, smpl/include/smpl/geometry/detail/shortcut.hpp:110-286)
 * and InterpolatePath (post_processing.cpp:464-523).  path: n points of nvars doubles.  The collision questions of the
 * greedy loops are answered from waypoint-parallel GPU batches.  stats (may be NULL): [0] edge batches, [1] configurations
 * checked.  out may be NULL to query the size. */
enum {
    SMPLX_PP_SHORTCUT = 1,          /* shortcut_path */
    SMPLX_PP_INTERPOLATE = 2,       /* interpolate_path */
    SMPLX_PP_UPSTREAM_LIMITS = 4    /* use upstream's limit test in CollisionSpace::interpolatePath; the default is the
                                       fork's (collision_space.cpp:592-597), under which interpolation of an in-limits
                                       path reports failure and leaves the path as it was */
};
int smplx_post_process_path(smplx_space* s, const double* path, int n, int flags, double* out, int cap, int* nout,
                            int64_t* stats);

#ifdef __cplusplus
}
#endif

#endif /* SMPL_AMD_H */
