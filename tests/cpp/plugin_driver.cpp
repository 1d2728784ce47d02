// tests/cpp/plugin_driver.cpp -- exercises include/smpl_amd/plugin.hpp the way smpl's own code drives its
// plugins: setGoal -> setStart -> repeated GetSuccs / GetGoalHeuristic / isStateToStateValid through the
// abstract base classes.  Prints one line per call; tests/test_gpu_plugin_cpp.py compares with the oracle.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include <smpl_amd/plugin.hpp>

using namespace smpl_amd;

static std::string slurp(const std::string& p)
{
    std::ifstream f(p);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    const std::string dir = argv[1];
    const std::string robot = slurp(dir + "/robot.txt"), mprim = slurp(dir + "/mprim.txt");
    std::ifstream q(dir + "/query.txt");
    double origin[3], res, max_dist;
    int n[3], nv;
    smplx_params P = {};
    q >> origin[0] >> origin[1] >> origin[2] >> n[0] >> n[1] >> n[2] >> res >> max_dist >> nv;
    for (int i = 0; i < nv; ++i) q >> P.resolutions[i];
    q >> P.bfs_inflation_radius >> P.cost_per_cell >> P.use_short_dist_mprims >> P.short_dist_mprims_thresh >>
        P.use_xyzrpy_snap_mprim >> P.xyzrpy_snap_dist_thresh >> P.xy_rotate_by_var3 >> P.use_long_and_short;
    RobotState start(nv), goal(nv), tol(nv);
    for (double& v : start) q >> v;
    for (double& v : goal) q >> v;
    for (double& v : tol) q >> v;
    int nexpand = 0;
    q >> nexpand;
    std::vector<int32_t> d2((size_t)n[0] * n[1] * n[2]);
    std::ifstream g(dir + "/grid.bin", std::ios::binary);
    g.read((char*)d2.data(), (std::streamsize)(d2.size() * sizeof(int32_t)));

    GpuPlanningContext ctx(robot, mprim, origin, n[0], n[1], n[2], res, max_dist, d2.data(), P);
    GpuManipLattice lattice(&ctx);
    GpuBfsHeuristic heur(&ctx);
    GpuCollisionChecker cc(&ctx);
    RobotPlanningSpace* space = &lattice;   // used through the abstract interfaces from here on
    RobotHeuristic* h = &heur;
    CollisionChecker* checker = cc.getExtension<CollisionChecker>();
    if (!checker || !lattice.getExtension<RobotPlanningSpace>() || lattice.getExtension<RobotHeuristic>()) return 3;

    if (!lattice.setGoalConfiguration(goal, tol)) return 4;
    if (!space->setStart(start)) return 5;
    printf("start %d goal %d valid %d edge %d\n", space->getStartStateID(), space->getGoalStateID(),
           (int)checker->isStateValid(start), (int)checker->isStateToStateValid(start, goal));
    std::vector<RobotState> path;
    checker->interpolatePath(start, goal, path);
    printf("interp %zu\n", path.size());
    std::vector<int> frontier{space->getStartStateID()};
    for (int i = 0; i < nexpand && i < (int)frontier.size(); ++i) {
        std::vector<int> succs, costs;
        space->GetSuccs(frontier[i], &succs, &costs);
        printf("succs %d :", frontier[i]);
        for (size_t k = 0; k < succs.size(); ++k) {
            printf(" %d/%d/%d", succs[k], costs[k], h->GetGoalHeuristic(succs[k]));
            bool seen = succs[k] == 0;
            for (int f : frontier) seen = seen || f == succs[k];
            if (!seen) frontier.push_back(succs[k]);
        }
        printf("\n");
    }
    std::vector<int> ids{space->getStartStateID()};
    std::vector<RobotState> states;
    if (!space->extractPath(ids, states) || states[0] != start) return 6;
    // PlannerInterface::postProcessPath through the mirror: shortcut + interpolate (upstream limit test) of a short path
    if (frontier.size() > 5) {
        std::vector<int> pid{frontier[0], frontier[1], frontier[5]};
        std::vector<RobotState> pp;
        if (!space->extractPath(pid, pp)) return 7;
        if (!PostProcessPath(&ctx, pp, true, true, true)) return 8;
        printf("post %zu", pp.size());
        for (double v : pp.back()) printf(" %.17g", v);
        for (double v : pp[pp.size() / 2]) printf(" %.17g", v);
        printf("\n");
    }
    printf("done\n");
    return 0;
}
