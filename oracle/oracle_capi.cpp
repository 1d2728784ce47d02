// oracle/oracle_capi.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
// C-ABI (ctypes) front-end of the CPU restatement in smpl_oracle.hpp.  Used by
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
#include "smpl_oracle.hpp"

#include <chrono>

using namespace oracle;

namespace {

struct Ctx {
    RobotDesc desc;
    RobotCollisionModel rcm;
    OccupancyGrid grid;
    std::unique_ptr<CollisionSpace> cc;
    PlanningRobotModel robot;
    BfsHeuristic heur;
    ActionSpace actions;
    ManipLattice lattice;
    std::unique_ptr<ARAStar> search;
    std::vector<double> resolutions;
    std::string error;
    std::vector<SuccessorRecord> trace;
};

void fill_dense(Ctx* c, unsigned char* flags, int* coord, double* q, int* h, int* cost, int* lookups)
{
    const int M = (int)c->actions.mprims.size();
    const int N = c->robot.jointVariableCount();
    for (int p = 0; p < M; ++p) {
        flags[p] = 0x10;
        h[p] = 0; cost[p] = 0; lookups[p] = 0;
        for (int i = 0; i < N; ++i) { coord[p * N + i] = 0; q[p * N + i] = 0.0; }
    }
    for (const SuccessorRecord& r : c->trace) {
        const int p = r.prim;
        flags[p] = (unsigned char)r.flags;
        h[p] = r.h;
        cost[p] = r.cost;
        lookups[p] = (int)r.lookups;
        for (size_t i = 0; i < r.q.size(); ++i) q[p * N + i] = r.q[i];
        for (size_t i = 0; i < r.coord.size(); ++i) coord[p * N + i] = r.coord[i];
    }
}

}  // namespace

extern "C" {

const char* orc_last_error(void* h) { return ((Ctx*)h)->error.c_str(); }

// resolutions: per planning variable (rad).  d2: nx*ny*nz squared cell distances (x-major, z fastest).
void* orc_create(const char* robot_text, const char* mprim_text, const double* origin, int nx, int ny, int nz,
                 double res, double max_dist, const int* d2, const double* resolutions, int nres,
                 double bfs_radius, int cost_per_cell, int use_short, double short_thresh,
                 int use_xyzrpy_snap, double xyzrpy_thresh, int xy_rotate_by_var3, int use_long_and_short)
{
    Ctx* c = new Ctx;
    std::string err;
    if (!parse_robot(robot_text, c->desc, &err)) { c->error = err; return c; }
    if (!build_collision_model(c->desc, c->rcm, &err)) { c->error = err; return c; }
    c->grid.init(origin, nx, ny, nz, res, max_dist, d2);
    c->cc.reset(new CollisionSpace(&c->grid, &c->rcm, c->desc.planning_joints));
    if (!c->robot.init(&c->rcm, c->desc.planning_joints, c->desc.planning_link)) { c->error = "robot init failed"; return c; }
    if (nres != c->robot.jointVariableCount()) { c->error = "resolution count"; return c; }
    c->resolutions.assign(resolutions, resolutions + nres);
    c->heur.init(&c->grid, bfs_radius, cost_per_cell);
    c->actions.clear();
    c->actions.params.enabled[SHORT_DISTANCE] = use_short != 0;
    c->actions.params.thresh[SHORT_DISTANCE] = short_thresh;
    c->actions.params.enabled[SNAP_TO_XYZ_RPY] = use_xyzrpy_snap != 0;
    c->actions.params.thresh[SNAP_TO_XYZ_RPY] = xyzrpy_thresh;
    c->actions.params.xy_rotate_by_var3 = xy_rotate_by_var3 != 0;
    c->actions.params.use_long_and_short = use_long_and_short != 0;
    if (!c->actions.load(mprim_text, c->resolutions, &err)) { c->error = "mprim: " + err; return c; }
    if (!c->lattice.init(&c->robot, c->cc.get(), &c->heur, &c->actions, c->resolutions)) { c->error = "lattice init"; return c; }
    c->search.reset(new ARAStar(&c->lattice));
    return c;
}

void orc_destroy(void* h) { delete (Ctx*)h; }

int orc_num_vars(void* h) { return ((Ctx*)h)->robot.jointVariableCount(); }
int orc_num_prims(void* h) { return (int)((Ctx*)h)->actions.mprims.size(); }
void orc_set_traversal_order(void* h, int order) { ((Ctx*)h)->cc->order = (TraversalOrder)order; }
void orc_set_padding(void* h, double padding) { ((Ctx*)h)->cc->padding = padding; }   // SelfCollisionModel m_padding

// --- compiled-model inspection (compared bit for bit with the product's host compiler) ---
int orc_model_counts(void* h, int* njoints, int* ntrees, int* nnodes, int* npairs)
{
    Ctx* c = (Ctx*)h;
    *njoints = (int)c->rcm.joint_names.size();
    *ntrees = (int)c->rcm.group_spheres_models.size();
    int n = 0;
    for (int sm : c->rcm.group_spheres_models) n += (int)c->rcm.spheres_models[sm].nodes.size();
    *nnodes = n;
    *npairs = (int)c->cc->checked_pairs.size();
    return 0;
}
// per joint: origin[12], k (motion-sphere factor)
void orc_model_joints(void* h, double* origins, double* k)
{
    Ctx* c = (Ctx*)h;
    for (size_t j = 0; j < c->rcm.joint_names.size(); ++j) {
        for (int i = 0; i < 3; ++i) for (int cc = 0; cc < 4; ++cc) origins[j * 12 + i * 4 + cc] = c->rcm.joint_origins[j].m[i][cc];
        k[j] = c->cc->rmcm.k[j];
    }
}
// nodes of all group trees concatenated in group order: center xyz + radius, left/right (tree-local, -1 leaf), link index
void orc_model_nodes(void* h, double* xyzr, int* left, int* right, int* link, int* tree_first)
{
    Ctx* c = (Ctx*)h;
    int o = 0, t = 0;
    for (int sm : c->rcm.group_spheres_models) {
        tree_first[t++] = o;
        for (const SphereNode& n : c->rcm.spheres_models[sm].nodes) {
            xyzr[o * 4 + 0] = n.center.x; xyzr[o * 4 + 1] = n.center.y; xyzr[o * 4 + 2] = n.center.z; xyzr[o * 4 + 3] = n.radius;
            left[o] = n.left; right[o] = n.right; link[o] = c->rcm.spheres_model_link[sm];
            ++o;
        }
    }
    tree_first[t] = o;
}
void orc_model_pairs(void* h, int* pairs)
{
    Ctx* c = (Ctx*)h;
    // pairs as indices into the group tree list
    auto pos = [&](int sm) {
        for (size_t i = 0; i < c->rcm.group_spheres_models.size(); ++i) if (c->rcm.group_spheres_models[i] == sm) return (int)i;
        return -1;
    };
    for (size_t i = 0; i < c->cc->checked_pairs.size(); ++i) {
        pairs[2 * i] = pos(c->cc->checked_pairs[i].first);
        pairs[2 * i + 1] = pos(c->cc->checked_pairs[i].second);
    }
}
void orc_discretization(void* h, int* vals, double* deltas)
{
    Ctx* c = (Ctx*)h;
    for (size_t i = 0; i < c->lattice.coord_vals.size(); ++i) { vals[i] = c->lattice.coord_vals[i]; deltas[i] = c->lattice.coord_deltas[i]; }
}

// --- primitives of the path ---
void orc_sincos(double x, double* s, double* c) { det_sincos(x, s, c); }
double orc_normalize_angle(double a) { return normalize_angle(a); }
void orc_state_to_coord(void* h, const double* q, int* coord)
{
    Ctx* c = (Ctx*)h;
    const int N = c->robot.jointVariableCount();
    std::vector<int> cc;
    c->lattice.stateToCoord(std::vector<double>(q, q + N), cc);
    std::copy(cc.begin(), cc.end(), coord);
}
int orc_check_joint_limits(void* h, const double* q)
{
    Ctx* c = (Ctx*)h;
    return c->robot.checkJointLimits(std::vector<double>(q, q + c->robot.jointVariableCount())) ? 1 : 0;
}
void orc_planning_fk(void* h, const double* q, double* xyz)
{
    Ctx* c = (Ctx*)h;
    c->robot.computePlanningLinkFK(std::vector<double>(q, q + c->robot.jointVariableCount()), xyz);
}
// world positions of all group tree nodes (same order as orc_model_nodes)
void orc_sphere_positions(void* h, const double* q, double* xyz)
{
    Ctx* c = (Ctx*)h;
    const int N = c->robot.jointVariableCount();
    for (int i = 0; i < N; ++i) c->cc->joint_vars[c->cc->planning_to_var[i]] = q[i];
    c->cc->rcs.setJointVarPositions(c->cc->joint_vars.data());
    int o = 0;
    for (int sm : c->rcm.group_spheres_models) {
        for (size_t n = 0; n < c->rcm.spheres_models[sm].nodes.size(); ++n) {
            const Vec3 p = c->cc->rcs.spherePos(sm, (int)n);
            xyz[o * 3] = p.x; xyz[o * 3 + 1] = p.y; xyz[o * 3 + 2] = p.z;
            ++o;
        }
    }
}
double orc_grid_sqdist(void* h, double x, double y, double z) { return ((Ctx*)h)->grid.getSquaredDist(x, y, z); }
void orc_world_to_grid(void* h, double x, double y, double z, int* cell)
{
    ((Ctx*)h)->grid.worldToGrid(x, y, z, cell[0], cell[1], cell[2]);
}
int orc_state_valid(void* h, const double* q, int* lookups)
{
    Ctx* c = (Ctx*)h;
    const long l0 = c->grid.lookups;
    const bool ok = c->cc->isStateValid(std::vector<double>(q, q + c->robot.jointVariableCount()));
    if (lookups) *lookups = (int)(c->grid.lookups - l0);
    return ok ? 1 : 0;
}
// n states one after another (the loop of benchmark_cc.cpp:234-256), timed; lookups may be null
void orc_state_valid_batch_timed(void* h, const double* Q, int n, unsigned char* out, int* lookups, double* seconds)
{
    Ctx* c = (Ctx*)h;
    const int N = c->robot.jointVariableCount();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) {
        int l = 0;
        out[i] = (unsigned char)orc_state_valid(h, Q + (size_t)i * N, &l);
        if (lookups) lookups[i] = l;
    }
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
// timing variant: look distances up in the reference's 48-byte array-of-structures cells (same values)
void orc_use_aos_cells(void* h, int on)
{
    Ctx* c = (Ctx*)h;
    if (on && c->grid.aos.empty()) c->grid.buildAosCells();
    c->grid.use_aos = on != 0;
}
int orc_waypoint_count(void* h, const double* a, const double* b)
{
    Ctx* c = (Ctx*)h;
    const int N = c->robot.jointVariableCount();
    return c->cc->waypointCount(std::vector<double>(a, a + N), std::vector<double>(b, b + N));
}
int orc_edge_valid(void* h, const double* a, const double* b, int* lookups)
{
    Ctx* c = (Ctx*)h;
    const int N = c->robot.jointVariableCount();
    const long l0 = c->grid.lookups;
    const bool ok = c->cc->isStateToStateValid(std::vector<double>(a, a + N), std::vector<double>(b, b + N));
    if (lookups) *lookups = (int)(c->grid.lookups - l0);
    return ok ? 1 : 0;
}
void orc_edge_valid_batch(void* h, const double* a, const double* b, int n, unsigned char* out, int* lookups)
{
    Ctx* c = (Ctx*)h;
    const int N = c->robot.jointVariableCount();
    for (int i = 0; i < n; ++i) {
        int l = 0;
        out[i] = (unsigned char)orc_edge_valid(h, a + (size_t)i * N, b + (size_t)i * N, &l);
        if (lookups) lookups[i] = l;
    }
    (void)c;
}

// --- heuristic ---
int orc_set_goal_joint(void* h, const double* angles, const double* tol)
{
    Ctx* c = (Ctx*)h;
    const int N = c->robot.jointVariableCount();
    GoalConstraint g;
    g.type = JOINT_STATE_GOAL;
    g.angles.assign(angles, angles + N);
    g.angle_tolerances.assign(tol, tol + N);
    double p[3];
    c->robot.computePlanningLinkFK(g.angles, p);  // planner_interface.cpp:1232-1235
    g.tgt_off_pose[0] = p[0]; g.tgt_off_pose[1] = p[1]; g.tgt_off_pose[2] = p[2];
    c->lattice.setGoal(g);
    c->search->set_goal(c->lattice.goal_state_id);
    return 1;
}
int orc_set_goal_xyz(void* h, const double* xyz, const double* tol)
{
    Ctx* c = (Ctx*)h;
    GoalConstraint g;
    g.type = XYZ_GOAL;
    for (int i = 0; i < 3; ++i) { g.tgt_off_pose[i] = xyz[i]; g.xyz_tolerance[i] = tol[i]; }
    c->lattice.setGoal(g);
    c->search->set_goal(c->lattice.goal_state_id);
    return 1;
}
void orc_goal_pose(void* h, double* xyz)
{
    Ctx* c = (Ctx*)h;
    for (int i = 0; i < 3; ++i) xyz[i] = c->lattice.goal.tgt_off_pose[i];
}
// padded (nx+2)(ny+2)(nz+2) BFS distance grid, reference node order (bfs3d.h:213-220)
long orc_bfs_size(void* h) { return (long)((Ctx*)h)->heur.bfs->dim_xyz; }
void orc_bfs_copy(void* h, int* out)
{
    Ctx* c = (Ctx*)h;
    std::copy(c->heur.bfs->dist.begin(), c->heur.bfs->dist.end(), out);
}
int orc_heuristic_q(void* h, const double* q)
{
    Ctx* c = (Ctx*)h;
    double p[3];
    c->robot.computePlanningLinkFK(std::vector<double>(q, q + c->robot.jointVariableCount()), p);
    return c->heur.heuristicAtPoint(p);
}
double orc_metric_goal_distance(void* h, double x, double y, double z) { return ((Ctx*)h)->heur.getMetricGoalDistance(x, y, z); }

// --- lattice ---
int orc_set_start(void* h, const double* q)
{
    Ctx* c = (Ctx*)h;
    if (!c->lattice.setStart(std::vector<double>(q, q + c->robot.jointVariableCount()))) return -1;
    c->search->set_start(c->lattice.start_state_id);
    return c->lattice.start_state_id;
}
int orc_num_states(void* h) { return (int)((Ctx*)h)->lattice.state_coords.size(); }
void orc_get_state(void* h, int id, double* q, int* coord)
{
    Ctx* c = (Ctx*)h;
    std::copy(c->lattice.state_angles[id].begin(), c->lattice.state_angles[id].end(), q);
    std::copy(c->lattice.state_coords[id].begin(), c->lattice.state_coords[id].end(), coord);
}
// GetSuccs on a real state id (creates states); returns count
int orc_get_succs(void* h, int id, int* succs, int* costs, int cap)
{
    Ctx* c = (Ctx*)h;
    std::vector<int> s, k;
    c->lattice.GetSuccs(id, &s, &k);
    const int n = std::min((int)s.size(), cap);
    for (int i = 0; i < n; ++i) { succs[i] = s[i]; costs[i] = k[i]; }
    return (int)s.size();
}
// evaluate all primitives of an arbitrary parent configuration without
// leaving anything in the state table; dense per-primitive outputs
void orc_eval_state(void* h, const double* q, unsigned char* flags, int* coord, double* sq, int* hh, int* cost, int* lookups)
{
    Ctx* c = (Ctx*)h;
    ManipLattice& L = c->lattice;
    const int N = c->robot.jointVariableCount();
    const size_t before = L.state_coords.size();
    const long ev = L.succ_evals, ex = L.expansions;
    const int tmp = L.reserveHashEntry();
    L.state_angles[tmp].assign(q, q + N);
    L.trace = &c->trace;
    std::vector<int> s, k;
    L.GetSuccs(tmp, &s, &k);
    L.trace = nullptr;
    fill_dense(c, flags, coord, sq, hh, cost, lookups);
    for (size_t id = before; id < L.state_coords.size(); ++id) {
        if (!L.state_coords[id].empty()) L.state_to_id.erase(L.state_coords[id]);
    }
    L.state_coords.resize(before);
    L.state_angles.resize(before);
    L.succ_evals = ev;
    L.expansions = ex;
}

// CPU baseline: the GetSuccs loop body over a batch of parent states, repeated until at least
// min_seconds of work has been done.  Returns evaluations (state x active primitive) and seconds.
void orc_eval_batch_timed(void* h, const double* Q, int B, double min_seconds, long* evals, long* valid, double* seconds,
                          int* passes)
{
    Ctx* c = (Ctx*)h;
    ManipLattice& L = c->lattice;
    const int N = c->robot.jointVariableCount();
    const long ev0 = L.succ_evals;
    long nvalid = 0;
    int np = 0;
    const auto t0 = std::chrono::steady_clock::now();
    double el = 0.0;
    do {
        for (int i = 0; i < B; ++i) {
            const size_t before = L.state_coords.size();
            const int tmp = L.reserveHashEntry();
            L.state_angles[tmp].assign(Q + (size_t)i * N, Q + (size_t)(i + 1) * N);
            std::vector<int> s, k;
            L.GetSuccs(tmp, &s, &k);
            nvalid += (long)s.size();
            for (size_t id = before; id < L.state_coords.size(); ++id)
                if (!L.state_coords[id].empty()) L.state_to_id.erase(L.state_coords[id]);
            L.state_coords.resize(before);
            L.state_angles.resize(before);
        }
        ++np;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < min_seconds);
    *evals = L.succ_evals - ev0;
    *valid = nvalid;
    *seconds = el;
    *passes = np;
    L.succ_evals = ev0;
}

// --- search ---
void orc_search_params(void* h, double eps0, double eps_final, double eps_delta, int improve, int bounded,
                       int max_exp_init, int max_exp)
{
    Ctx* c = (Ctx*)h;
    c->search->initial_eps = eps0;
    c->search->final_eps = std::max(eps_final, 1.0);
    c->search->delta_eps = eps_delta;
    c->search->improve = improve != 0;
    c->search->bounded = bounded != 0;
    c->search->max_expansions_init = max_exp_init;
    c->search->max_expansions = max_exp;
}
// returns 1 on success; path ids -> path_ids (cap), *path_len, *cost
int orc_plan(void* h, int* path_ids, int cap, int* path_len, int* cost, int* expansions, long* succ_evals,
             double* satisfied_eps, double* seconds)
{
    Ctx* c = (Ctx*)h;
    c->search->force_planning_from_scratch();
    c->search->expansion_log.clear();
    std::vector<int> sol;
    int co = 0;
    const long ev0 = c->lattice.succ_evals;
    const auto t0 = std::chrono::steady_clock::now();
    const int ok = c->search->replan(&sol, &co);
    const auto t1 = std::chrono::steady_clock::now();
    *seconds = std::chrono::duration<double>(t1 - t0).count();
    *path_len = (int)sol.size();
    for (int i = 0; i < std::min((int)sol.size(), cap); ++i) path_ids[i] = sol[i];
    *cost = co;
    *expansions = c->search->expand_count;
    *succ_evals = c->lattice.succ_evals - ev0;
    *satisfied_eps = c->search->satisfied_eps;
    return ok;
}
int orc_expansion_log_size(void* h) { return (int)((Ctx*)h)->search->expansion_log.size(); }
void orc_expansion_log(void* h, int* out)
{
    Ctx* c = (Ctx*)h;
    std::copy(c->search->expansion_log.begin(), c->search->expansion_log.end(), out);
}
long orc_total_lookups(void* h) { return ((Ctx*)h)->grid.lookups; }

// --- path post-processing (N3).  mode bits: 1 shortcut, 2 interpolate, 4 upstream limit test (fork bug off)
int orc_post_process(void* h, const double* path, int n, int mode, double* out, int cap, long* edge_checks, long* state_checks)
{
    Ctx* c = (Ctx*)h;
    const int N = c->robot.jointVariableCount();
    std::vector<std::vector<double>> p(n);
    for (int i = 0; i < n; ++i) p[i].assign(path + (size_t)i * N, path + (size_t)(i + 1) * N);
    PostProcessor pp{&c->robot, c->cc.get()};
    pp.fork_interpolate_limits_bug = !(mode & 4);
    pp.postProcessPath(p, (mode & 1) != 0, (mode & 2) != 0);
    for (int i = 0; i < (int)p.size() && i < cap; ++i) std::copy(p[i].begin(), p[i].end(), out + (size_t)i * N);
    if (edge_checks) *edge_checks = pp.edge_checks;
    if (state_checks) *state_checks = pp.state_checks;
    return (int)p.size();
}

// the shortcut loop alone, answered from tables (same interface as oracle/shortcut_ref_driver.cpp): P points,
// cost[P*P], valid[P*P]; returns the number of output indices
int orc_shortcut_tables(int P, const double* cost, const int* valid, int* out)
{
    std::vector<double> seg;
    for (int i = 0; i + 1 < P; ++i) seg.push_back(cost[(size_t)i * P + i + 1]);
    std::vector<int> idx;
    shortcut_indices((size_t)P, seg, [&](size_t a, size_t b, double& c) {
        if (!valid[a * P + b]) return false;
        c = cost[a * P + b];
        return true;
    }, idx);
    std::copy(idx.begin(), idx.end(), out);
    return (int)idx.size();
}

// --- intrusive heap exerciser: ops[i] = {code, key}; codes 0 push(new elem with key), 1 pop,
// 2 decrease(elem index key>>20 to priority key&0xFFFFF), 3 erase(elem index key), 4 make (after
// rewriting all priorities p -> (p*7919+13)%1000), 5 increase(elem, priority).  Emits the element
// index at the top after every op (-1 if empty) into out.  Mirrors oracle/heap_ref_driver.cpp.
struct HElem { size_t heap_index = 0; int prio = 0; int idx = 0; };
struct HLess { bool operator()(const HElem& a, const HElem& b) const { return a.prio < b.prio; } };
void orc_heap_run(const int* ops, int nops, int* out)
{
    std::vector<std::unique_ptr<HElem>> elems;
    IntrusiveHeap<HElem, HLess> heap;
    for (int i = 0; i < nops; ++i) {
        const int code = ops[2 * i], key = ops[2 * i + 1];
        if (code == 0) {
            elems.emplace_back(new HElem);
            elems.back()->prio = key;
            elems.back()->idx = (int)elems.size() - 1;
            heap.push(elems.back().get());
        } else if (code == 1) {
            if (!heap.empty()) heap.pop();
        } else if (code == 2 || code == 5) {
            const int e = key >> 20, p = key & 0xFFFFF;
            if (e < (int)elems.size() && heap.contains(elems[e].get())) {
                elems[e]->prio = p;
                if (code == 2) heap.decrease(elems[e].get()); else heap.increase(elems[e].get());
            }
        } else if (code == 3) {
            if (key < (int)elems.size() && heap.contains(elems[key].get())) heap.erase(elems[key].get());
        } else if (code == 4) {
            for (size_t k = 1; k < heap.data.size(); ++k) heap.data[k]->prio = (heap.data[k]->prio * 7919 + 13) % 1000;
            heap.make();
        }
        out[i] = heap.empty() ? -1 : heap.min()->idx;
    }
}

}  // extern "C"
