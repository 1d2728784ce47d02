// include/smpl_amd/plugin.hpp -- C++ mirror of smpl's plugin interfaces for the expansion path,
// implemented over the C-ABI (include/smpl_amd.h).  Header-only; link with libsmpl_amd.so.
//
// Same class and method names, argument meaning and error behaviour (bool success, empty successor list
// = dead end) as the reference, minus Eigen/ROS/SBPL types:
//   CollisionChecker     smpl/include/smpl/collision_checker.h:48-130
//   RobotHeuristic       smpl/include/smpl/heuristic/robot_heuristic.h:53-101
//   RobotPlanningSpace   smpl/include/smpl/graph/robot_planning_space.h:60-218
//                        (+ SBPL DiscreteSpaceInformation::GetSuccs, Heuristic::GetGoalHeuristic)
//   Extension            smpl/include/smpl/extension.h:40-60
// In a real integration these classes derive from the reference's own bases; the bodies
// stay as they are here.
#pragma once

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <limits>
#include <stdexcept>
#include <string>
#include <typeinfo>
#include <vector>

#include "../smpl_amd.h"

namespace smpl_amd {

typedef std::vector<double> RobotState;   // smpl/include/smpl/types.h:66

// smpl/include/smpl/extension.h:40-60
class Extension {
public:
    virtual ~Extension() {}
    template <class T> T* getExtension() { return dynamic_cast<T*>(getExtension(typeid(T).hash_code())); }
    virtual Extension* getExtension(size_t class_code) = 0;
};
template <class T> inline size_t GetClassCode() { return typeid(T).hash_code(); }

// smpl/include/smpl/types.h:148-195
enum GoalType { INVALID_GOAL_TYPE = -1, XYZ_GOAL, XYZ_RPY_GOAL, JOINT_STATE_GOAL, NUMBER_OF_GOAL_TYPES };
enum GroupType { FAILURE = -2, ANY = -1, BASE = 0, ARM = 1, BASE_ISO = 2 };
struct GoalConstraint {
    RobotState angles;                       // joint-state goals
    std::vector<double> angle_tolerances;
    std::vector<double> pose;                // workspace goals: planning-link pose (x, y, z, R, P, Y)
    double xyz_offset[3] = {0, 0, 0};
    double xyz_tolerance[3] = {0, 0, 0};
    double rpy_tolerance[3] = {0, 0, 0};
    std::vector<double> tgt_off_pose;        // goal pose offset from the planning link
    int xyz[3] = {0, 0, 0};
    GoalType type = INVALID_GOAL_TYPE;
};

// smpl/include/smpl/collision_checker.h:48-130, every virtual of the fork
class CollisionChecker : public virtual Extension {
public:
    virtual bool isStateValid(const RobotState& state, bool verbose = false) = 0;                       // :62
    virtual bool isStateValid(const RobotState& state, double& distToObst, bool verbose = false) = 0;  // :64
    virtual bool isStateToStateValid(const RobotState& start, const RobotState& finish, bool verbose = false) = 0;   // :77-80
    // [FORK] :82-88 -- non-pure, and the reference's default body is empty (falls off the end); here it answers with
    // the three-argument form and leaves the two distances untouched
    virtual bool isStateToStateValid(const RobotState& angles0, const RobotState& angles1, double& distToObst, int& distToObstCells,
                                     bool verbose = false)
    {
        (void)distToObst; (void)distToObstCells;
        return isStateToStateValid(angles0, angles1, verbose);
    }
    virtual bool interpolatePath(const RobotState& start, const RobotState& finish, std::vector<RobotState>& path) = 0;   // :99-102
    // [FORK] no-op hooks (:110-128)
    virtual void setLastExpansionStep(int) {}
    virtual void markGridForExpandedState(const RobotState&, const RobotState&, int) {}
    virtual void resetCellsMarking(int) {}
    virtual void setClearanceThreshold(double) {}
};

class RobotPlanningSpace;

// smpl/include/smpl/graph/robot_planning_space_observer.h:42-50
class RobotPlanningSpaceObserver {
public:
    virtual ~RobotPlanningSpaceObserver() {}
    virtual void updateStart(const RobotState&) {}
    virtual void updateGoal(const GoalConstraint&) {}
};

// SBPL Heuristic + smpl/include/smpl/heuristic/robot_heuristic.h:53-101
class RobotHeuristic : public RobotPlanningSpaceObserver, public virtual Extension {
public:
    static const int Infinity = 32767;                       // :62 (INT16_MAX)
    bool init(RobotPlanningSpace* space) { m_space = space; return space != nullptr; }   // robot_heuristic.cpp:39-50
    RobotPlanningSpace* planningSpace() { return m_space; }
    virtual double getMetricStartDistance(double x, double y, double z) = 0;
    virtual double getMetricGoalDistance(double x, double y, double z) = 0;
    virtual bool setGoal(const GoalConstraint&) { return true; }
    virtual int GetGoalHeuristic(int state_id) = 0;
    virtual int GetGoalHeuristic(int state_id, int /*planning_group*/, int /*base_heuristic_idx*/) { return GetGoalHeuristic(state_id); }   // :88-91
    virtual int GetStartHeuristic(int state_id) = 0;
    virtual int GetFromToHeuristic(int from_id, int to_id) = 0;

private:
    RobotPlanningSpace* m_space = nullptr;
};

// smpl/include/smpl/robot_model.h:50-151, the part the lattice touches; the kinematics themselves live in the
// compiled model behind the C-ABI (smplx_model_create), so this carries names and limits only
class RobotModel : public virtual Extension {
public:
    virtual int jointVariableCount() const = 0;
    virtual bool checkJointLimits(const RobotState& state, bool verbose = false) = 0;
};

// smpl/include/smpl/planning_params.h:63-155: the fields this path reads; the rest of the reference's struct
// (log names, shortcutting options) stays with the caller
struct PlanningParams {
    smplx_params engine;        // discretization, cost_per_cell, planning_link_sphere_radius, primitive gating
    std::string mprim_text;     // contents of the file named by "mprim_filename"
};

// SBPL DiscreteSpaceInformation + smpl/include/smpl/graph/robot_planning_space.h:60-218
class RobotPlanningSpace : public virtual Extension {
public:
    virtual bool init(RobotModel* robot, CollisionChecker* checker, const PlanningParams* params)   // :68-71
    {
        m_robot = robot; m_checker = checker; m_params = params;
        return robot && checker && params;
    }
    virtual bool setStart(const RobotState& state) = 0;                                            // :73
    virtual bool setMultipleStart(const std::vector<RobotState>&) { return false; }               // [FORK] :74
    virtual bool setGoal(const GoalConstraint& goal) = 0;                                          // :76
    virtual int getStartStateID() const = 0;
    virtual int getGoalStateID() const = 0;
    virtual std::vector<int> getStartStatesID() const { return std::vector<int>(1, getStartStateID()); }   // [FORK] :80-83
    virtual bool extractPath(const std::vector<int>& ids, std::vector<RobotState>& path) = 0;      // :85-87
    virtual bool insertHeuristic(RobotHeuristic* h)                                                // :89, robot_planning_space.cpp:96-108
    {
        if (!h || hasHeuristic(h)) return false;
        m_heuristics.push_back(h);
        return true;
    }
    virtual bool eraseHeuristic(const RobotHeuristic* h)
    {
        for (size_t i = 0; i < m_heuristics.size(); ++i)
            if (m_heuristics[i] == h) { m_heuristics.erase(m_heuristics.begin() + i); return true; }
        return false;
    }
    virtual bool hasHeuristic(const RobotHeuristic* h)
    {
        for (RobotHeuristic* x : m_heuristics) if (x == h) return true;
        return false;
    }
    RobotModel* robot() { return m_robot; }
    CollisionChecker* collisionChecker() { return m_checker; }
    const PlanningParams* params() const { return m_params; }
    size_t numHeuristics() const { return m_heuristics.size(); }
    RobotHeuristic* heuristic(size_t i) { return i < m_heuristics.size() ? m_heuristics[i] : nullptr; }
    void insertObserver(RobotPlanningSpaceObserver* obs) { if (obs && !hasObserver(obs)) m_obs.push_back(obs); }
    void eraseObserver(RobotPlanningSpaceObserver* obs)
    {
        for (size_t i = 0; i < m_obs.size(); ++i) if (m_obs[i] == obs) { m_obs.erase(m_obs.begin() + i); return; }
    }
    bool hasObserver(RobotPlanningSpaceObserver* obs) const
    {
        for (RobotPlanningSpaceObserver* x : m_obs) if (x == obs) return true;
        return false;
    }
    void notifyStartChanged(const RobotState& state) { for (RobotPlanningSpaceObserver* o : m_obs) o->updateStart(state); }   // robot_planning_space.cpp:117-123
    void notifyGoalChanged(const GoalConstraint& goal) { for (RobotPlanningSpaceObserver* o : m_obs) o->updateGoal(goal); }

    // DiscreteSpaceInformation side
    virtual void GetSuccs(int state_id, std::vector<int>* succs, std::vector<int>* costs) = 0;     // :135-138
    virtual void GetPreds(int state_id, std::vector<int>* preds, std::vector<int>* costs) = 0;     // :172-175
    virtual void PrintState(int state_id, bool verbose, FILE* f = nullptr) = 0;                    // :177-180
    virtual int GetGoalHeuristic(int state_id)                                                    // robot_planning_space.cpp:133-140
    {
        return numHeuristics() == 0 ? 0 : heuristic(0)->GetGoalHeuristic(state_id);
    }
    virtual int GetStartHeuristic(int state_id) { return numHeuristics() == 0 ? 0 : heuristic(0)->GetStartHeuristic(state_id); }
    virtual int GetFromToHeuristic(int from_id, int to_id)
    {
        return numHeuristics() == 0 ? 0 : heuristic(0)->GetFromToHeuristic(from_id, to_id);
    }
    // [FORK] group / expansion-step entry points (:140-158), driven only by the fork's other planners.  With
    // group ANY they are GetSuccs (manip_lattice.cpp:315-411 differs from :219-313 only in the no-op
    // setLastExpansionStep hook); other groups are outside this path and yield no successors.
    virtual void GetSuccsByGroup(int state_id, std::vector<int>* succs, std::vector<int>* costs, std::vector<int>* /*clearance_cells*/, int group)
    {
        if (group == ANY) GetSuccs(state_id, succs, costs);
    }
    virtual void GetSuccsByGroupAndExpansion(int state_id, std::vector<int>* succs, std::vector<int>* costs, int group, int /*expansion_step*/)
    {
        if (group == ANY) GetSuccs(state_id, succs, costs);
    }
    virtual void GetSuccsWithExpansion(int state_id, std::vector<int>* succs, std::vector<int>* costs, int expansion_step)
    {
        if (m_checker) m_checker->setLastExpansionStep(expansion_step);
        GetSuccs(state_id, succs, costs);
    }
    virtual void GetPredsByGroupAndExpansion(int, std::vector<int>*, std::vector<int>*, std::vector<int>*, int, int, int) {}
    virtual bool updateMultipleStartStates(std::vector<int>*, std::vector<double>*, int) { return false; }
    virtual void setMotionPlanRequestType(int) {}
    virtual void setSelectedStartId(int) {}

private:
    RobotModel* m_robot = nullptr;
    CollisionChecker* m_checker = nullptr;
    const PlanningParams* m_params = nullptr;
    std::vector<RobotHeuristic*> m_heuristics;
    std::vector<RobotPlanningSpaceObserver*> m_obs;
};

// One query context on one GPU.  Owns the C-ABI handles; the three plugin facades below share it, the
// way the reference's lattice, heuristic and checker share one OccupancyGrid and one RobotModel.
class GpuPlanningContext {
public:
    GpuPlanningContext(const std::string& robot_text, const std::string& mprim_text, const double origin[3], int nx, int ny,
                       int nz, double res, double max_dist, const int32_t* d2, const smplx_params& params)
    {
        check(smplx_grid_create(origin, nx, ny, nz, res, max_dist, d2, &grid_));
        check(smplx_model_create(robot_text.c_str(), &model_));
        check(smplx_space_create(model_, grid_, mprim_text.c_str(), &params, &space_));
        nvars_ = smplx_space_num_vars(space_);
        nprims_ = smplx_space_num_prims(space_);
        res_ = res;
    }
    ~GpuPlanningContext()
    {
        smplx_space_destroy(space_);
        smplx_model_destroy(model_);
        smplx_grid_destroy(grid_);
    }
    GpuPlanningContext(const GpuPlanningContext&) = delete;
    GpuPlanningContext& operator=(const GpuPlanningContext&) = delete;
    smplx_space* space() const { return space_; }
    int nvars() const { return nvars_; }
    int nprims() const { return nprims_; }
    double resolution() const { return res_; }
    static void check(int code)
    {
        if (code != SMPLX_OK) throw std::runtime_error(std::string("smpl_amd: ") + smplx_last_error());
    }

private:
    smplx_grid* grid_ = nullptr;
    smplx_model* model_ = nullptr;
    smplx_space* space_ = nullptr;
    int nvars_ = 0, nprims_ = 0;
    double res_ = 0.0;
};

// RobotModel facade over the compiled model: variable count and KDLRobotModel::checkJointLimits
// (sbpl_kdl_robot_model/src/kdl_robot_model.cpp:210-235)
class GpuRobotModel : public RobotModel {
public:
    using Extension::getExtension;
    explicit GpuRobotModel(GpuPlanningContext* ctx) : ctx_(ctx) {}
    int jointVariableCount() const override { return ctx_->nvars(); }
    bool checkJointLimits(const RobotState& state, bool = false) override
    {
        uint8_t ok = 0;
        return (int)state.size() == ctx_->nvars() && smplx_check_joint_limits(ctx_->space(), state.data(), 1, &ok) == SMPLX_OK && ok;
    }
    Extension* getExtension(size_t class_code) override { return class_code == GetClassCode<RobotModel>() ? this : nullptr; }

private:
    GpuPlanningContext* ctx_;
};

// sbpl::collision::CollisionSpace (sbpl_collision_checking/include/sbpl_collision_checking/collision_space.h:66-266)
class GpuCollisionChecker : public CollisionChecker {
public:
    using Extension::getExtension;   // keep the typed lookup visible next to the override
    explicit GpuCollisionChecker(GpuPlanningContext* ctx) : ctx_(ctx) {}
    bool isStateValid(const RobotState& state, bool = false) override
    {
        uint8_t ok = 0;
        return smplx_cc_state_valid_batch(ctx_->space(), state.data(), 1, &ok, nullptr) == SMPLX_OK && ok;
    }
    // collision_space.h:202-205: the reference's override has an empty body; this one answers with the validity and
    // the value CollisionSpace::isStateValid starts its own `dist` at (collision_space.cpp:532-536)
    bool isStateValid(const RobotState& state, double& distToObst, bool verbose = false) override
    {
        distToObst = std::numeric_limits<double>::max();
        return isStateValid(state, verbose);
    }
    using CollisionChecker::isStateToStateValid;   // keep the fork's 5-argument overload visible
    bool isStateToStateValid(const RobotState& start, const RobotState& finish, bool = false) override
    {
        uint8_t ok = 0;
        return smplx_cc_edge_valid_batch(ctx_->space(), start.data(), finish.data(), 1, &ok, nullptr, nullptr) == SMPLX_OK && ok;
    }
    // batched forms: what a frontier-batched caller uses instead of one call per edge
    bool areStatesToStatesValid(const std::vector<double>& starts, const std::vector<double>& finishes, std::vector<uint8_t>& ok)
    {
        const int n = (int)(starts.size() / ctx_->nvars());
        ok.assign(n, 0);
        return smplx_cc_edge_valid_batch(ctx_->space(), starts.data(), finishes.data(), n, ok.data(), nullptr, nullptr) == SMPLX_OK;
    }
    bool interpolatePath(const RobotState& start, const RobotState& finish, std::vector<RobotState>& path) override
    {
        int n = 0;
        if (smplx_cc_interpolate(ctx_->space(), start.data(), finish.data(), nullptr, 0, &n) != SMPLX_OK) return false;
        std::vector<double> buf((size_t)n * ctx_->nvars());
        if (smplx_cc_interpolate(ctx_->space(), start.data(), finish.data(), buf.data(), n, &n) != SMPLX_OK) return false;
        path.clear();
        for (int i = 0; i < n; ++i) path.emplace_back(buf.begin() + (size_t)i * ctx_->nvars(), buf.begin() + (size_t)(i + 1) * ctx_->nvars());
        return true;
    }
    Extension* getExtension(size_t class_code) override
    {
        return class_code == GetClassCode<CollisionChecker>() ? this : nullptr;   // collision_space.cpp:523-529
    }

private:
    GpuPlanningContext* ctx_;
};

// sbpl::motion::ManipLattice (smpl/include/smpl/graph/manip_lattice.h:63-307)
class GpuManipLattice : public RobotPlanningSpace {
public:
    using Extension::getExtension;   // keep the typed lookup visible next to the override
    explicit GpuManipLattice(GpuPlanningContext* ctx) : ctx_(ctx) {}
    // GoalConstraint with JOINT_STATE_GOAL / XYZ_GOAL (manip_lattice.cpp:1982-1997); observers (the BFS
    // heuristic) are notified inside the engine: BFS_3D::run completes before this returns
    bool setGoalConfiguration(const RobotState& angles, const RobotState& tolerances)
    {
        return smplx_set_goal_joint(ctx_->space(), angles.data(), tolerances.data()) == SMPLX_OK;
    }
    bool setGoalPosition(const double xyz[3], const double tol[3]) { return smplx_set_goal_xyz(ctx_->space(), xyz, tol) == SMPLX_OK; }
    // ManipLattice::setGoal (manip_lattice.cpp:1982-1997): dispatch on the goal type, then notify the observers
    // (the heuristic; its BFS has already been run to completion inside the engine)
    bool setGoal(const GoalConstraint& goal) override
    {
        bool ok = false;
        switch (goal.type) {
        case XYZ_GOAL: {
            const std::vector<double>& p = goal.tgt_off_pose.size() >= 3 ? goal.tgt_off_pose : goal.pose;
            if (p.size() < 3) return false;
            const double xyz[3] = {p[0], p[1], p[2]};
            ok = setGoalPosition(xyz, goal.xyz_tolerance);
        } break;
        case JOINT_STATE_GOAL:
            if ((int)goal.angles.size() != ctx_->nvars() || goal.angle_tolerances.size() != goal.angles.size()) return false;
            ok = setGoalConfiguration(goal.angles, goal.angle_tolerances);
            break;
        default:
            return false;   // XYZ_RPY_GOAL needs IK (out of this path's scope)
        }
        if (ok) notifyGoalChanged(goal);
        return ok;
    }
    bool setStart(const RobotState& state) override
    {
        if (smplx_set_start(ctx_->space(), state.data(), nullptr) != SMPLX_OK) return false;
        notifyStartChanged(state);
        return true;
    }
    int getStartStateID() const override { return smplx_start_id(ctx_->space()); }
    int getGoalStateID() const override { return smplx_goal_id(ctx_->space()); }
    void GetSuccs(int state_id, std::vector<int>* succs, std::vector<int>* costs) override
    {
        const int cap = ctx_->nprims();
        std::vector<int32_t> s(cap), c(cap);
        int n = 0;
        if (smplx_get_succs(ctx_->space(), state_id, s.data(), c.data(), cap, &n) != SMPLX_OK) {
            // GetSuccs cannot report: the list stays empty (a dead end to the search), the error stays on the space
            // (smplx_space_status / engineStatus()) and is logged once
            if (!logged_) { fprintf(stderr, "smpl_amd: GetSuccs(%d) failed: %s\n", state_id, smplx_last_error()); logged_ = true; }
            return;
        }
        succs->insert(succs->end(), s.begin(), s.begin() + n);
        costs->insert(costs->end(), c.begin(), c.begin() + n);
    }
    void GetPreds(int, std::vector<int>*, std::vector<int>*) override {}   // "GetPreds unimplemented" (manip_lattice.cpp:1238-1241)
    // manip_lattice.cpp:152-197
    void PrintState(int state_id, bool /*verbose*/, FILE* f = nullptr) override
    {
        if (!f) f = stdout;
        if (state_id == getGoalStateID()) { fprintf(f, "<goal state>\n"); return; }
        RobotState q(ctx_->nvars());
        if (smplx_get_state(ctx_->space(), state_id, q.data(), nullptr) != SMPLX_OK) { fprintf(f, "<unknown state %d>\n", state_id); return; }
        fprintf(f, "{ ");
        for (size_t i = 0; i < q.size(); ++i) fprintf(f, "%.3g%s", q[i], i + 1 < q.size() ? ", " : "");
        fprintf(f, " }\n");
    }
    // the first engine error a GetSuccs call swallowed (SMPLX_OK if none)
    int engineStatus(std::string* msg = nullptr) const
    {
        char buf[512];
        const int st = smplx_space_status(ctx_->space(), buf, (int)sizeof(buf));
        if (msg) *msg = buf;
        return st;
    }
    // optional, not in the reference: ids the search will expand soon ride along in the next frontier batch
    void hintFrontier(const std::vector<int>& ids) { smplx_hint_frontier(ctx_->space(), ids.data(), (int)ids.size()); }
    bool extractPath(const std::vector<int>& ids, std::vector<RobotState>& path) override
    {
        std::vector<double> q(ids.size() * (size_t)ctx_->nvars());
        if (smplx_extract_path(ctx_->space(), ids.data(), (int)ids.size(), q.data()) != SMPLX_OK) return false;
        path.clear();
        for (size_t i = 0; i < ids.size(); ++i) path.emplace_back(q.begin() + i * ctx_->nvars(), q.begin() + (i + 1) * ctx_->nvars());
        return true;
    }
    Extension* getExtension(size_t class_code) override
    {
        return class_code == GetClassCode<RobotPlanningSpace>() ? this : nullptr;   // manip_lattice.cpp:2157-2170
    }

private:
    GpuPlanningContext* ctx_;
    bool logged_ = false;
};

// sbpl::motion::BfsHeuristic (smpl/include/smpl/heuristic/bfs_heuristic.h:49-105)
class GpuBfsHeuristic : public RobotHeuristic {
public:
    using Extension::getExtension;   // keep the typed lookup visible next to the override
    explicit GpuBfsHeuristic(GpuPlanningContext* ctx) : ctx_(ctx) {}
    int GetGoalHeuristic(int state_id) override
    {
        int32_t h = 0;
        return smplx_get_goal_heuristic(ctx_->space(), state_id, &h) == SMPLX_OK ? h : 0;
    }
    int GetStartHeuristic(int) override { return 0; }         // "unimplemented" in the reference (bfs_heuristic.cpp:165-169)
    int GetFromToHeuristic(int from_id, int to_id) override                             // bfs_heuristic.cpp:171-180
    {
        return to_id == smplx_goal_id(ctx_->space()) ? GetGoalHeuristic(from_id) : 0;
    }
    // bfs_heuristic.cpp:129-138 / 103-127, served by the engine (the same values its primitive gating uses)
    double getMetricGoalDistance(double x, double y, double z) override
    {
        const double p[3] = {x, y, z};
        double d = 0.0;
        return smplx_bfs_metric_goal_distance(ctx_->space(), p, 1, &d) == SMPLX_OK ? d : 0.0;
    }
    double getMetricStartDistance(double x, double y, double z) override
    {
        const double p[3] = {x, y, z};
        double d = 0.0;
        return smplx_bfs_metric_start_distance(ctx_->space(), p, 1, &d) == SMPLX_OK ? d : 0.0;
    }
    Extension* getExtension(size_t class_code) override
    {
        return class_code == GetClassCode<RobotHeuristic>() ? this : nullptr;       // bfs_heuristic.cpp:140-146
    }

private:
    GpuPlanningContext* ctx_;
};

// sbpl::motion::PlannerInterface::postProcessPath (smpl_ros/src/ros/planner_interface.cpp:2651-2697): ShortcutPath with
// ShortcutType::JOINT_SPACE (smpl/include/smpl/post_processing.h:49-71) between two InterpolatePath passes.  The greedy
// loops are the reference's; their collision questions are answered from waypoint-parallel GPU batches.
// upstream_limit_test = false keeps the fork's CollisionSpace::interpolatePath limit test (collision_space.cpp:592-597).
inline bool PostProcessPath(GpuPlanningContext* ctx, std::vector<RobotState>& path, bool shortcut_path, bool interpolate_path,
                            bool upstream_limit_test = false)
{
    const int N = ctx->nvars();
    std::vector<double> in(path.size() * (size_t)N);
    for (size_t i = 0; i < path.size(); ++i) std::copy(path[i].begin(), path[i].end(), in.begin() + i * N);
    const int flags = (shortcut_path ? SMPLX_PP_SHORTCUT : 0) | (interpolate_path ? SMPLX_PP_INTERPOLATE : 0) |
                      (upstream_limit_test ? SMPLX_PP_UPSTREAM_LIMITS : 0);
    int n = 0;
    if (smplx_post_process_path(ctx->space(), in.data(), (int)path.size(), flags, nullptr, 0, &n, nullptr) != SMPLX_OK) return false;
    std::vector<double> out((size_t)std::max(n, 1) * N);
    if (smplx_post_process_path(ctx->space(), in.data(), (int)path.size(), flags, out.data(), n, &n, nullptr) != SMPLX_OK) return false;
    path.clear();
    for (int i = 0; i < n; ++i) path.emplace_back(out.begin() + (size_t)i * N, out.begin() + (size_t)(i + 1) * N);
    return true;
}

}  // namespace smpl_amd
