// tests/cpp/sbpl_loop_driver.cpp -- an SBPL-shaped planner loop over include/smpl_amd/plugin.hpp.
//
// The caller here knows ONLY what smpl's ARAStar knows of its environment (smpl/src/search/arastar.cpp:531-568, 613-627):
// RobotPlanningSpace::GetSuccs(id, &succs, &costs) and GetGoalHeuristic(id) through the abstract base classes.  It
// never calls smplx_hint_frontier or any other engine-specific entry point, so every GetSuccs miss is served by the
// engine's own hint-free speculation (include/smpl_amd.h, smplx_hint_frontier note).  The loop restates ARA* with
// the reference's heap sift rules (smpl/include/smpl/detail/intrusive_heap.hpp:346-395) so that the expansion log can
// be compared with the oracle's, id for id.  Test infrastructure (tests/test_gpu_plugin_cpp.py); also timed by bench.py.
//
// usage: sbpl_loop_driver <dir> [log]      dir holds robot.txt mprim.txt grid.bin query.txt (see the python test)
// prints: "result <solved> <cost> <expansions> <path_len> <seconds>", "stats ...", optionally "log id id id ..."
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

#include <smpl_amd/plugin.hpp>

using namespace smpl_amd;

namespace {

const unsigned kInf = 1000000000u;   // SBPL INFINITECOST

struct Node {
    unsigned g = kInf, h = 0, f = kInf, eg = kInf;
    unsigned short iteration_closed = 0, call_number = 0;
    int bp = -1, heap_index = 0;
    bool incons = false;
};

// the planner: what arastar.cpp does with a DiscreteSpaceInformation, nothing more
class PlainARAStar {
public:
    PlainARAStar(RobotPlanningSpace* space) : space_(space) {}
    double eps0 = 5.0, eps_final = 1.0, eps_delta = 1.0;
    int max_expansions_init = 0, max_expansions = 0;   // TimeParameters::EXPANSIONS bounds
    std::vector<int> log;
    int expansions = 0;
    double satisfied_eps = std::numeric_limits<double>::infinity();

    bool replan(int start_id, int goal_id, std::vector<int>* path, int* cost)
    {
        start_ = start_id; goal_ = goal_id;
        heap_.assign(1, 0);
        incons_.clear();
        ++call_;
        reinit(start_); reinit(goal_);
        n_[start_].g = 0;
        n_[start_].f = key(n_[start_]);
        push(start_);
        iteration_ = 1;
        eps_ = eps0;
        int num = 0, err = 0;
        const double fin = eps_final < 1.0 ? 1.0 : eps_final;
        while (satisfied_eps > fin) {
            if (eps_ == satisfied_eps) {
                ++iteration_;
                eps_ -= eps_delta;
                if (eps_ < fin) eps_ = fin;
                for (int s : incons_) { n_[s].incons = false; push(s); }
                for (size_t i = 1; i < heap_.size(); ++i) n_[heap_[i]].f = key(n_[heap_[i]]);
                for (size_t i = (heap_.size() - 1) >> 1; i >= 1; --i) down(i);
                incons_.clear();
            }
            err = improve(num);
            if (err) break;
            satisfied_eps = eps_;
        }
        expansions = num;
        if (satisfied_eps == std::numeric_limits<double>::infinity()) return false;
        path->clear();
        for (int s = goal_; s >= 0; s = n_[s].bp) path->insert(path->begin(), s);
        *cost = (int)n_[goal_].g;
        return true;
    }

private:
    RobotPlanningSpace* space_;
    std::vector<Node> n_;
    std::vector<int> heap_, incons_;
    double eps_ = 1.0;
    int iteration_ = 1, call_ = 0, start_ = -1, goal_ = 0;

    bool less(int a, int b) const { return n_[a].f < n_[b].f; }
    void down(size_t pivot)
    {
        if (pivot >= heap_.size()) return;
        size_t left = pivot << 1, right = left + 1;
        const int tmp = heap_[pivot];
        while (left < heap_.size()) {
            size_t c = right;
            if (right >= heap_.size() || less(heap_[left], heap_[right])) c = left;
            if (!less(heap_[c], tmp)) break;
            heap_[pivot] = heap_[c];
            n_[heap_[pivot]].heap_index = (int)pivot;
            pivot = c;
            left = pivot << 1; right = left + 1;
        }
        heap_[pivot] = tmp;
        n_[tmp].heap_index = (int)pivot;
    }
    void up(size_t pivot)
    {
        const int tmp = heap_[pivot];
        while (pivot != 1) {
            const size_t p = pivot >> 1;
            if (less(heap_[p], tmp)) break;
            heap_[pivot] = heap_[p];
            n_[heap_[pivot]].heap_index = (int)pivot;
            pivot = p;
        }
        heap_[pivot] = tmp;
        n_[tmp].heap_index = (int)pivot;
    }
    void push(int e) { n_[e].heap_index = (int)heap_.size(); heap_.push_back(e); up(heap_.size() - 1); }
    void pop()
    {
        n_[heap_[1]].heap_index = 0;
        heap_[1] = heap_.back();
        heap_.pop_back();
        down(1);
    }
    unsigned key(const Node& s) const { return s.g + (unsigned)(long long)(eps_ * s.h); }
    void reinit(int id)
    {
        if ((int)n_.size() <= id) n_.resize(id + 1);
        Node& s = n_[id];
        if (s.call_number != (unsigned short)call_) {
            s.g = kInf;
            s.h = (unsigned)space_->GetGoalHeuristic(id);    // the ONLY heuristic access
            s.f = kInf; s.eg = kInf;
            s.iteration_closed = 0;
            s.call_number = (unsigned short)call_;
            s.bp = -1;
            s.incons = false;
        }
    }
    bool timed_out(int elapsed) const
    {
        if (max_expansions_init <= 0) return false;
        if (satisfied_eps == std::numeric_limits<double>::infinity()) return elapsed >= max_expansions_init;
        return elapsed >= max_expansions;
    }
    int improve(int& elapsed)
    {
        std::vector<int> succs, costs;
        while (heap_.size() > 1) {
            const int m = heap_[1];
            if (n_[m].f >= n_[goal_].f || m == goal_) return 0;
            if (timed_out(elapsed)) return 4;
            pop();
            n_[m].iteration_closed = (unsigned short)iteration_;
            n_[m].eg = n_[m].g;
            log.push_back(m);
            succs.clear(); costs.clear();
            space_->GetSuccs(m, &succs, &costs);             // the ONLY successor access
            const unsigned eg = n_[m].eg;
            for (size_t i = 0; i < succs.size(); ++i) {
                const int nid = succs[i];
                reinit(nid);
                Node& t = n_[nid];
                const int new_cost = (int)(eg + (unsigned)costs[i]);
                if ((unsigned)new_cost < t.g) {
                    t.g = (unsigned)new_cost;
                    t.bp = m;
                    if (t.iteration_closed != (unsigned short)iteration_) {
                        t.f = key(t);
                        if (t.heap_index != 0) up(t.heap_index);
                        else push(nid);
                    } else if (!t.incons) {
                        incons_.push_back(nid);
                    }
                }
            }
            ++elapsed;
        }
        return 5;
    }
};

std::string slurp(const std::string& p)
{
    std::ifstream f(p);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

}  // namespace

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    const std::string dir = argv[1];
    const bool want_log = argc > 2 && std::strcmp(argv[2], "log") == 0;
    const std::string robot = slurp(dir + "/robot.txt"), mprim = slurp(dir + "/mprim.txt");
    std::ifstream q(dir + "/query.txt");
    double origin[3], res, max_dist;
    int n[3], nv;
    PlanningParams params;
    smplx_params& P = params.engine;
    P = smplx_params();
    q >> origin[0] >> origin[1] >> origin[2] >> n[0] >> n[1] >> n[2] >> res >> max_dist >> nv;
    for (int i = 0; i < nv; ++i) q >> P.resolutions[i];
    q >> P.bfs_inflation_radius >> P.cost_per_cell >> P.use_short_dist_mprims >> P.short_dist_mprims_thresh >>
        P.use_xyzrpy_snap_mprim >> P.xyzrpy_snap_dist_thresh >> P.xy_rotate_by_var3 >> P.use_long_and_short;
    GoalConstraint goal;
    goal.type = JOINT_STATE_GOAL;
    RobotState start(nv);
    goal.angles.resize(nv); goal.angle_tolerances.resize(nv);
    for (double& v : start) q >> v;
    for (double& v : goal.angles) q >> v;
    for (double& v : goal.angle_tolerances) q >> v;
    double eps0, eps_final, eps_delta;
    int max_init, max_rep;
    q >> eps0 >> eps_final >> eps_delta >> max_init >> max_rep;
    if (!q) { fprintf(stderr, "bad query.txt\n"); return 2; }
    std::vector<int32_t> d2((size_t)n[0] * n[1] * n[2]);
    std::ifstream g(dir + "/grid.bin", std::ios::binary);
    g.read((char*)d2.data(), (std::streamsize)(d2.size() * sizeof(int32_t)));
    params.mprim_text = mprim;

    GpuPlanningContext ctx(robot, mprim, origin, n[0], n[1], n[2], res, max_dist, d2.data(), P);
    GpuRobotModel model(&ctx);
    GpuCollisionChecker cc(&ctx);
    GpuManipLattice lattice(&ctx);
    GpuBfsHeuristic heur(&ctx);
    // the wiring PlannerInterface does (smpl_ros/src/ros/planner_interface.cpp:131-230, 424-470):
    // space.init(robot, checker, params); heuristic.init(space); space.insertHeuristic(heuristic)
    RobotPlanningSpace* space = &lattice;
    if (!space->init(&model, &cc, &params)) return 3;
    if (!heur.init(space)) return 3;
    space->insertObserver(&heur);
    if (!space->insertHeuristic(&heur)) return 3;
    if (!model.checkJointLimits(start)) return 4;
    if (!space->setGoal(goal)) return 4;
    if (!space->setStart(start)) return 5;

    PlainARAStar planner(space);
    planner.eps0 = eps0; planner.eps_final = eps_final; planner.eps_delta = eps_delta;
    planner.max_expansions_init = max_init; planner.max_expansions = max_rep;
    std::vector<int> path;
    int cost = 0;
    const auto t0 = std::chrono::steady_clock::now();
    const bool ok = planner.replan(space->getStartStateID(), space->getGoalStateID(), &path, &cost);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::string msg;
    if (lattice.engineStatus(&msg) != SMPLX_OK) { fprintf(stderr, "engine error: %s\n", msg.c_str()); return 6; }
    printf("result %d %d %d %zu %.6f %.17g\n", ok ? 1 : 0, ok ? cost : 0, planner.expansions, ok ? path.size() : (size_t)0, secs,
           planner.satisfied_eps);
    int64_t ctr[6] = {0, 0, 0, 0, 0, 0};
    smplx_space_counters(ctx.space(), ctr);
    printf("stats gpu_batches=%lld cache_hits=%lld cache_misses=%lld committed_succ_evals=%lld gpu_succ_evals_total=%lld states=%lld\n",
           (long long)ctr[0], (long long)ctr[1], (long long)ctr[2], (long long)ctr[3], (long long)ctr[4], (long long)ctr[5]);
    // metric distances through the heuristic mirror (RobotHeuristic virtuals), goal pose first
    double gp[3];
    smplx_goal_pose(ctx.space(), gp);
    printf("metric %.17g %.17g\n", heur.getMetricGoalDistance(gp[0], gp[1], gp[2]), heur.getMetricStartDistance(gp[0], gp[1], gp[2]));
    if (ok) {
        printf("path");
        for (int id : path) printf(" %d", id);
        printf("\n");
    }
    if (want_log) {
        printf("log");
        for (int id : planner.log) printf(" %d", id);
        printf("\n");
    }
    lattice.PrintState(space->getStartStateID(), false, stdout);
    printf("done\n");
    return 0;
}
