// smpl_amd/csrc/model_compile.h -- host-side "model compiler": plain-text robot description ->
// the flat arrays the kernels read (SmplxModelDev).  It does, for the plain-text format, what
// RobotCollisionModel / CollisionSphereModelTree / RobotMotionCollisionModel do for URDF + YAML
// (sbpl_collision_checking/src/robot_collision_model.cpp:117-623, base_collision_models.cpp:337-444,
//  robot_motion_collision_model.cpp:41-275).
#pragma once

#include <string>
#include <vector>

#include "device_types.h"

namespace smplx {

struct HostModel {
    SmplxModelDev dev;                        // thresholds (nodes[].thr) are filled when bound to a grid
    std::vector<std::string> var_names;       // planning variables, in planning order
    std::vector<std::string> joint_names;     // depth-first order, as in dev.joints
    std::vector<double> joint_k;              // motion-sphere factor per joint (depth-first order)
    std::vector<int> file_joint_index;        // depth-first position -> index in the file's joint list
    std::string planning_link;
    std::string error;
};

// returns false and sets m.error on malformed input or when a limit of device_types.h is exceeded
bool compile_robot_text(const char* text, HostModel& m);

// smallest squared cell distance i in [0, dmax_sqrd] with (res*sqrt(i))^2 >= (r+pad)^2, else dmax_sqrd+1:
// the integer form of CheckSphereCollision (collision_operations.h:67-77) over the sqrt table of
// distance_map.hpp:140-144
int sphere_threshold(double radius, double padding, double res, int dmax_sqrd);

// largest i in [0, dmax_sqrd] with res*sqrt(i) <= radius, -1 if none (bfs_heuristic.cpp:343)
int wall_threshold(double radius, double res, int dmax_sqrd);

struct HostActions {
    SmplxActionsDev dev;
    std::string error;
};
// .mprim loader (manip_lattice_action_space.cpp:103-261): upstream rows (cols == nvars) or fork rows
// (cols == nvars + 2: deltas, group, weight)
bool load_mprim_text(const char* text, const double* resolutions, int nvars, HostActions& a);

// packed model for LDS staging (device_types.h SMPLX_BH_*); returns the byte count, 0 if it does not fit cap
size_t pack_model_blob(const SmplxModelDev& m, unsigned char* out, size_t cap);

// C++ text defining the chain structure of the model as compile-time constants (CM_* arrays): the input of the
// per-robot specialisation of the collision kernel (kernels.hip, SMPLX_CONST_MODEL)
std::string model_const_header(const SmplxModelDev& m);

// ManipLattice::init discretisation (manip_lattice.cpp:125-139)
void fill_discretization(SmplxModelDev& m, const double* resolutions);

}  // namespace smplx
