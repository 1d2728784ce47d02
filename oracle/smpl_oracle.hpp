// oracle/smpl_oracle.hpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement (C++17, no dependencies) of the ARA* state-expansion hot path
// of dyouakim/smpl: ManipLattice successor generation, BFS-3D heuristic and the
// sbpl_collision_checking sphere-tree vs. voxel-grid check, plus the ARA* caller
// (it defines expansion order and therefore state ids).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// include, link or execute this.  The product (smpl_amd/) never does.
//
// PARITY STATUS: "parity unpinned" for everything except the intrusive heap and
// the path shortcutting loop.  The reference cannot be built here (Eigen, Boost,
// SBPL, ROS, KDL, urdf are absent; smpl/config.h is cmake-generated), and its own
// tests hold no numeric golden vectors for this path.  The
// two header-only, std-only pieces -- smpl/include/smpl/intrusive_heap.h and
// smpl/include/smpl/geometry/shortcut.h -- are compiled in place by oracle/Makefile
// into oracle/_ref/ and pin IntrusiveHeap and shortcut_indices below
// (tests/golden/heap_ref.json, tests/golden/shortcut_ref.json).  Everything else
// follows the cited reference lines and is pinned by hand-derived known-answer
// tests (tests/test_oracle_kat.py).
//
// Third-party arithmetic that is NOT in /root/reference (orocos-kdl FK, Eigen
// products, libm sin/cos) is restated with the "arithmetic contract" of
// DESIGN.md section 3 (fixed expression order, no FMA contraction, a fixed
// polynomial sincos) so that this file and the HIP kernels are comparable
// bit for bit.  Define ORACLE_USE_LIBM_SINCOS to swap in libm's sin/cos.
//
// All file:line citations are relative to /root/reference.
#pragma once

#include <algorithm>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace oracle {

// ---------------------------------------------------------------------------
// deterministic math (DESIGN.md section 3 "arithmetic contract")
// ---------------------------------------------------------------------------

// sin and cos by 3-term Cody-Waite reduction to [-pi/4, pi/4] and the classic
// degree-13/14 minimax kernels.  Every operation is a single IEEE-754 double
// +,-,* (no FMA), so the same sequence gives the same bits on x86-64 and gfx950.
inline void det_sincos(double x, double* s_out, double* c_out)
{
#ifdef ORACLE_USE_LIBM_SINCOS
    *s_out = std::sin(x);
    *c_out = std::cos(x);
#else
    const double INVPIO2 = 6.36619772367581382433e-01;
    const double P1 = 0x1.921fb54400000p+0;   // pi/2, leading 33 bits
    const double P2 = 0x1.0b4611a600000p-34;  // next 33 bits
    const double P3 = 0x1.3198a2e000000p-69;  // next 33 bits
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double fn = std::rint(x * INVPIO2);
    const int n = (int)fn;
    const double r = ((x - fn * P1) - fn * P2) - fn * P3;
    const double z = r * r;
    // sin kernel
    const double v = z * r;
    const double rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    const double ks = r + v * (S1 + z * rs);
    // cos kernel
    const double rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double kc = w + (((1.0 - w) - hz) + z * rc);
    switch (n & 3) {
    case 0: *s_out = ks; *c_out = kc; break;
    case 1: *s_out = kc; *c_out = -ks; break;
    case 2: *s_out = -ks; *c_out = -kc; break;
    default: *s_out = -kc; *c_out = ks; break;
    }
#endif
}

// smpl/include/smpl/angles.h:45-71
inline double normalize_angle(double angle)
{
    if (std::fabs(angle) > 2.0 * M_PI) {
        angle = std::fmod(angle, 2.0 * M_PI);
    }
    if (angle < -M_PI) {
        angle += 2.0 * M_PI;
    }
    if (angle > M_PI) {
        angle -= 2.0 * M_PI;
    }
    return angle;
}

inline double normalize_angle_positive(double angle)
{
    angle = normalize_angle(angle);
    if (angle < 0.0) {
        angle += 2.0 * M_PI;
    }
    return angle;
}

// smpl/include/smpl/angles.h:88-100
inline double shortest_angle_diff(double af, double ai) { return normalize_angle(af - ai); }
inline double shortest_angle_dist(double af, double ai) { return std::fabs(shortest_angle_diff(af, ai)); }

struct Vec3 { double x = 0, y = 0, z = 0; };

inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(double s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
// Eigen squaredNorm / norm of a 3-vector: ((x*x + y*y) + z*z)
inline double sqnorm(Vec3 a) { return (a.x * a.x + a.y * a.y) + a.z * a.z; }
inline double norm(Vec3 a) { return std::sqrt(sqnorm(a)); }

// 3x4 affine transform, row-major; last row is implicitly [0 0 0 1]
struct Affine {
    double m[3][4];
    static Affine Identity()
    {
        Affine a;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) a.m[i][j] = (i == j) ? 1.0 : 0.0;
        return a;
    }
};

// T = A * B for affine transforms (Eigen Transform<double,3,Affine> product
// as used at sbpl_collision_checking/include/sbpl_collision_checking/robot_collision_state.h:419-421).
// Arithmetic contract: linear(i,j) = (a_i0*b_0j + a_i1*b_1j) + a_i2*b_2j;
//                      trans(i)    = ((a_i0*t0 + a_i1*t1) + a_i2*t2) + a_i3.
inline Affine mul(const Affine& a, const Affine& b)
{
    Affine r;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            r.m[i][j] = (a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j]) + a.m[i][2] * b.m[2][j];
        }
        r.m[i][3] = ((a.m[i][0] * b.m[0][3] + a.m[i][1] * b.m[1][3]) + a.m[i][2] * b.m[2][3]) + a.m[i][3];
    }
    return r;
}

// p = T * c  (robot_collision_state.h:576)
inline Vec3 apply(const Affine& t, Vec3 c)
{
    Vec3 p;
    p.x = ((t.m[0][0] * c.x + t.m[0][1] * c.y) + t.m[0][2] * c.z) + t.m[0][3];
    p.y = ((t.m[1][0] * c.x + t.m[1][1] * c.y) + t.m[1][2] * c.z) + t.m[1][3];
    p.z = ((t.m[2][0] * c.x + t.m[2][1] * c.y) + t.m[2][2] * c.z) + t.m[2][3];
    return p;
}

// Rotation matrix of Eigen::AngleAxisd(angle, axis).toRotationMatrix() (Eigen3,
// third party, restated): used by the general-axis joint transform
// (sbpl_collision_checking/src/transform_functions.h:96-102) and the motion
// sphere sampling (robot_motion_collision_model.cpp:170-173).
inline void angle_axis_matrix(double angle, Vec3 axis, double R[3][3])
{
    double s, c;
    det_sincos(angle, &s, &c);
    const Vec3 sin_axis = s * axis;
    const Vec3 cos1_axis = (1.0 - c) * axis;
    double tmp;
    tmp = cos1_axis.x * axis.y;
    R[0][1] = tmp - sin_axis.z;
    R[1][0] = tmp + sin_axis.z;
    tmp = cos1_axis.x * axis.z;
    R[0][2] = tmp + sin_axis.y;
    R[2][0] = tmp - sin_axis.y;
    tmp = cos1_axis.y * axis.z;
    R[1][2] = tmp - sin_axis.x;
    R[2][1] = tmp + sin_axis.x;
    R[0][0] = cos1_axis.x * axis.x + c;
    R[1][1] = cos1_axis.y * axis.y + c;
    R[2][2] = cos1_axis.z * axis.z + c;
}

// ---------------------------------------------------------------------------
// robot description (plain text; replaces URDF + collision-model YAML, which
// need urdf/XmlRpc -- SURVEY.md section 2.2 item 24)
// ---------------------------------------------------------------------------

enum JointType { FIXED = 0, REVOLUTE = 1, CONTINUOUS = 2, PRISMATIC = 3 };

struct JointDesc {
    std::string name;
    JointType type = FIXED;
    std::string parent, child;
    double oxyz[3] = {0, 0, 0};
    double orpy[3] = {0, 0, 0};
    double axis[3] = {0, 0, 1};
    double lo = 0, hi = 0;
};

struct SphereDesc {
    std::string link, name;
    double x = 0, y = 0, z = 0, radius = 0;
    int priority = 1;
};

struct RobotDesc {
    std::string name;
    std::vector<std::string> links;
    std::vector<JointDesc> joints;
    std::vector<SphereDesc> spheres;
    std::string group_name;
    std::vector<std::string> group_links;
    std::vector<std::pair<std::string, std::string>> acm_allowed;
    std::vector<std::string> planning_joints;
    std::string planning_link;
};

inline bool parse_robot(const std::string& text, RobotDesc& out, std::string* err = nullptr)
{
    std::istringstream in(text);
    std::string line;
    int lineno = 0;
    auto fail = [&](const std::string& m) {
        if (err) *err = "line " + std::to_string(lineno) + ": " + m;
        return false;
    };
    while (std::getline(in, line)) {
        ++lineno;
        auto hash = line.find('#');
        if (hash != std::string::npos) line.resize(hash);
        std::istringstream ls(line);
        std::string kw;
        if (!(ls >> kw)) continue;
        if (kw == "robot") {
            ls >> out.name;
        } else if (kw == "link") {
            std::string n;
            if (!(ls >> n)) return fail("link needs a name");
            out.links.push_back(n);
        } else if (kw == "joint") {
            JointDesc j;
            std::string type;
            if (!(ls >> j.name >> type >> j.parent >> j.child)) return fail("bad joint");
            if (type == "fixed") j.type = FIXED;
            else if (type == "revolute") j.type = REVOLUTE;
            else if (type == "continuous") j.type = CONTINUOUS;
            else if (type == "prismatic") j.type = PRISMATIC;
            else return fail("unknown joint type " + type);
            if (!(ls >> j.oxyz[0] >> j.oxyz[1] >> j.oxyz[2] >> j.orpy[0] >> j.orpy[1] >> j.orpy[2] >>
                  j.axis[0] >> j.axis[1] >> j.axis[2] >> j.lo >> j.hi)) return fail("bad joint numbers");
            out.joints.push_back(j);
        } else if (kw == "sphere") {
            SphereDesc s;
            if (!(ls >> s.link >> s.name >> s.x >> s.y >> s.z >> s.radius >> s.priority)) return fail("bad sphere");
            out.spheres.push_back(s);
        } else if (kw == "group") {
            ls >> out.group_name;
            std::string l;
            while (ls >> l) out.group_links.push_back(l);
        } else if (kw == "acm") {
            std::string a, b;
            if (!(ls >> a >> b)) return fail("bad acm");
            out.acm_allowed.emplace_back(a, b);
        } else if (kw == "planning_joints") {
            std::string j;
            while (ls >> j) out.planning_joints.push_back(j);
        } else if (kw == "planning_link") {
            ls >> out.planning_link;
        } else {
            return fail("unknown keyword " + kw);
        }
    }
    if (out.links.empty()) return fail("no links");
    return true;
}

// ---------------------------------------------------------------------------
// sphere tree (sbpl_collision_checking/src/base_collision_models.cpp:337-444,
// 569-641).  Node array is in post-order: children before parents, root last
// (base_collision_models.h:109 root() == back()).
// ---------------------------------------------------------------------------

struct SphereNode {
    Vec3 center;
    double radius = 0;
    int left = -1, right = -1;  // -1,-1 for leaves
    int leaf_id = -1;           // index into the link's sphere list for leaves
    bool isLeaf() const { return left == right; }
};

struct SphereTree {
    std::vector<SphereNode> nodes;
    int root() const { return (int)nodes.size() - 1; }
};

struct LeafRef { double x, y, z, r; int id; };

// libstdc++ std::partition for bidirectional iterators (the reference calls
// std::partition at base_collision_models.cpp:394-400; the element order it
// leaves is implementation-defined, so the GNU algorithm is restated to make
// the tree shape a function of the input only).
template <class It, class Pred>
inline It gnu_partition(It first, It last, Pred pred)
{
    while (true) {
        while (true) {
            if (first == last) return first;
            else if (pred(*first)) ++first;
            else break;
        }
        --last;
        while (true) {
            if (first == last) return first;
            else if (!pred(*last)) --last;
            else break;
        }
        std::iter_swap(first, last);
        ++first;
    }
}

// base_collision_models.cpp:569-592
inline void optimal_bounding_sphere(const SphereNode& s1, const SphereNode& s2, Vec3& c, double& r)
{
    const Vec3 p = s1.center;
    const Vec3 q = s2.center;
    const Vec3 v = q - p;
    const double dist = norm(v);
    if (s1.radius > dist + s2.radius) {
        c = s1.center;
        r = s1.radius;
    } else if (s2.radius > dist + s1.radius) {
        c = s2.center;
        r = s2.radius;
    } else {
        // Eigen normalized(): v / norm  (division, element-wise)
        const Vec3 vn = {v.x / dist, v.y / dist, v.z / dist};
        const Vec3 a = q + s2.radius * vn;  // q + vn * r2 (scalar product commutes exactly)
        const Vec3 b = p - s1.radius * vn;
        c = 0.5 * (a + b);
        r = 0.5 * norm(a - b);
    }
}

// base_collision_models.cpp:594-641
inline int largest_bbox_axis(const LeafRef* first, const LeafRef* last)
{
    if (first == last) return 0;
    double mn[3] = {first->x, first->y, first->z};
    double mx[3] = {first->x, first->y, first->z};
    for (const LeafRef* it = first; it != last; ++it) {
        mn[0] = std::min(mn[0], it->x); mn[1] = std::min(mn[1], it->y); mn[2] = std::min(mn[2], it->z);
        mx[0] = std::max(mx[0], it->x); mx[1] = std::max(mx[1], it->y); mx[2] = std::max(mx[2], it->z);
    }
    const double spanx = mx[0] - mn[0], spany = mx[1] - mn[1], spanz = mx[2] - mn[2];
    if (spanx > spany && spanx > spanz) return 0;
    else if (spany > spanz) return 1;
    else return 2;
}

// base_collision_models.cpp:337-444 (buildRecursive)
inline int build_sphere_tree_rec(SphereTree& tree, LeafRef* first, LeafRef* last)
{
    if (first == last) return -1;
    const auto count = last - first;
    if (count == 1) {
        SphereNode n;
        n.center = {first->x, first->y, first->z};
        n.radius = first->r;
        n.leaf_id = first->id;
        tree.nodes.push_back(n);
        return (int)tree.nodes.size() - 1;
    }
    const int split_axis = largest_bbox_axis(first, last);
    Vec3 cc;
    for (LeafRef* it = first; it != last; ++it) cc = cc + Vec3{it->x, it->y, it->z};
    cc = {cc.x / (double)count, cc.y / (double)count, cc.z / (double)count};
    double cr = 0.0;
    for (LeafRef* it = first; it != last; ++it) {
        const double radius = norm(Vec3{it->x, it->y, it->z} - cc) + it->r;
        if (radius > cr) cr = radius;
    }
    LeafRef* mid;
    if (split_axis == 0) mid = gnu_partition(first, last, [&](const LeafRef& s) { return s.x < cc.x; });
    else if (split_axis == 1) mid = gnu_partition(first, last, [&](const LeafRef& s) { return s.y < cc.y; });
    else mid = gnu_partition(first, last, [&](const LeafRef& s) { return s.z < cc.z; });
    if (first == mid || mid == last) mid = first + (count >> 1);
    const int li = build_sphere_tree_rec(tree, first, mid);
    const int ri = build_sphere_tree_rec(tree, mid, last);
    Vec3 gc;
    double gr;
    optimal_bounding_sphere(tree.nodes[li], tree.nodes[ri], gc, gr);
    SphereNode n;
    if (gr < cr) { n.center = gc; n.radius = gr; }
    else { n.center = cc; n.radius = cr; }
    n.left = li;
    n.right = ri;
    tree.nodes.push_back(n);
    return (int)tree.nodes.size() - 1;
}

inline SphereTree build_sphere_tree(std::vector<LeafRef> leaves)
{
    SphereTree t;
    if (!leaves.empty()) build_sphere_tree_rec(t, leaves.data(), leaves.data() + leaves.size());
    return t;
}

// ---------------------------------------------------------------------------
// RobotCollisionModel: flat arrays (sbpl_collision_checking/src/robot_collision_model.cpp:117-623)
// ---------------------------------------------------------------------------

enum TransformKind { TK_FIXED = 0, TK_REV_X, TK_REV_Y, TK_REV_Z, TK_REV_GENERIC, TK_PRISMATIC };

struct RobotCollisionModel {
    std::vector<std::string> link_names;
    std::vector<int> link_parent_joint;               // -1 for root
    std::vector<std::vector<int>> link_child_joints;
    std::vector<std::string> joint_names;
    std::vector<JointType> joint_types;
    std::vector<TransformKind> joint_kinds;
    std::vector<int> joint_parent_link, joint_child_link;
    std::vector<Affine> joint_origins;
    std::vector<Vec3> joint_axes;
    std::vector<int> joint_var;                       // var index or -1
    std::vector<std::string> var_names;
    std::vector<int> var_joint;
    std::vector<double> var_min, var_max;
    std::vector<bool> var_continuous, var_bounded;
    // spheres models: one per link that has spheres
    std::vector<int> link_spheres_model;              // -1 if none
    std::vector<SphereTree> spheres_models;
    std::vector<int> spheres_model_link;
    std::vector<std::vector<SphereDesc>> spheres_model_leaves;
    // group
    std::vector<int> group_links;
    std::vector<int> group_spheres_models;            // in group link order
    // allowed link pairs (by link index), symmetric
    std::vector<std::vector<bool>> acm_allowed;

    int linkIndex(const std::string& n) const
    {
        for (size_t i = 0; i < link_names.size(); ++i) if (link_names[i] == n) return (int)i;
        return -1;
    }
    int jointIndex(const std::string& n) const
    {
        for (size_t i = 0; i < joint_names.size(); ++i) if (joint_names[i] == n) return (int)i;
        return -1;
    }
    int varIndex(const std::string& n) const
    {
        for (size_t i = 0; i < var_names.size(); ++i) if (var_names[i] == n) return (int)i;
        return -1;
    }
};

// origin = Translation(xyz) * Rz(yaw) * Ry(pitch) * Rx(roll)  (urdf rpy convention)
inline Affine origin_from_xyz_rpy(const double xyz[3], const double rpy[3])
{
    double sr, cr, sp, cp, sy, cy;
    det_sincos(rpy[0], &sr, &cr);
    det_sincos(rpy[1], &sp, &cp);
    det_sincos(rpy[2], &sy, &cy);
    Affine a;
    a.m[0][0] = cy * cp;  a.m[0][1] = (cy * sp) * sr - sy * cr;  a.m[0][2] = (cy * sp) * cr + sy * sr;
    a.m[1][0] = sy * cp;  a.m[1][1] = (sy * sp) * sr + cy * cr;  a.m[1][2] = (sy * sp) * cr - cy * sr;
    a.m[2][0] = -sp;      a.m[2][1] = cp * sr;                   a.m[2][2] = cp * cr;
    a.m[0][3] = xyz[0]; a.m[1][3] = xyz[1]; a.m[2][3] = xyz[2];
    return a;
}

inline bool build_collision_model(const RobotDesc& d, RobotCollisionModel& m, std::string* err = nullptr)
{
    auto fail = [&](const std::string& s) { if (err) *err = s; return false; };
    m = RobotCollisionModel();
    m.link_names = d.links;
    const int nl = (int)d.links.size();
    m.link_parent_joint.assign(nl, -1);
    m.link_child_joints.assign(nl, {});
    for (const JointDesc& j : d.joints) {
        const int jidx = (int)m.joint_names.size();
        const int pl = m.linkIndex(j.parent), cl = m.linkIndex(j.child);
        if (pl < 0 || cl < 0) return fail("joint " + j.name + " references unknown link");
        m.joint_names.push_back(j.name);
        m.joint_types.push_back(j.type);
        m.joint_parent_link.push_back(pl);
        m.joint_child_link.push_back(cl);
        m.joint_origins.push_back(origin_from_xyz_rpy(j.oxyz, j.orpy));
        m.joint_axes.push_back({j.axis[0], j.axis[1], j.axis[2]});
        m.link_parent_joint[cl] = jidx;
        m.link_child_joints[pl].push_back(jidx);
        // robot_collision_model.cpp:331-407: transform function by exact axis
        TransformKind k = TK_FIXED;
        if (j.type == REVOLUTE || j.type == CONTINUOUS) {
            if (j.axis[0] == 1.0 && j.axis[1] == 0.0 && j.axis[2] == 0.0) k = TK_REV_X;
            else if (j.axis[0] == 0.0 && j.axis[1] == 1.0 && j.axis[2] == 0.0) k = TK_REV_Y;
            else if (j.axis[0] == 0.0 && j.axis[1] == 0.0 && j.axis[2] == 1.0) k = TK_REV_Z;
            else k = TK_REV_GENERIC;
        } else if (j.type == PRISMATIC) {
            k = TK_PRISMATIC;
        }
        m.joint_kinds.push_back(k);
        if (j.type == FIXED) {
            m.joint_var.push_back(-1);
        } else {
            m.joint_var.push_back((int)m.var_names.size());
            m.var_names.push_back(j.name);
            m.var_joint.push_back(jidx);
            if (j.type == CONTINUOUS) {
                m.var_continuous.push_back(true);
                m.var_bounded.push_back(false);
                m.var_min.push_back(-std::numeric_limits<double>::infinity());
                m.var_max.push_back(std::numeric_limits<double>::infinity());
            } else {
                m.var_continuous.push_back(false);
                m.var_bounded.push_back(true);
                m.var_min.push_back(j.lo);
                m.var_max.push_back(j.hi);
            }
        }
    }
    m.link_spheres_model.assign(nl, -1);
    for (int l = 0; l < nl; ++l) {
        std::vector<SphereDesc> mine;
        for (const SphereDesc& s : d.spheres) if (s.link == d.links[l]) mine.push_back(s);
        if (mine.empty()) continue;
        std::vector<LeafRef> refs;
        for (size_t i = 0; i < mine.size(); ++i) refs.push_back({mine[i].x, mine[i].y, mine[i].z, mine[i].radius, (int)i});
        m.link_spheres_model[l] = (int)m.spheres_models.size();
        m.spheres_models.push_back(build_sphere_tree(refs));
        m.spheres_model_link.push_back(l);
        m.spheres_model_leaves.push_back(mine);
    }
    for (const SphereDesc& s : d.spheres) if (m.linkIndex(s.link) < 0) return fail("sphere on unknown link " + s.link);
    for (const std::string& gl : d.group_links) {
        const int l = m.linkIndex(gl);
        if (l < 0) return fail("group link unknown: " + gl);
        m.group_links.push_back(l);
        if (m.link_spheres_model[l] >= 0) m.group_spheres_models.push_back(m.link_spheres_model[l]);
    }
    // self_collision_model.cpp:280-312: adjacent links are allowed
    m.acm_allowed.assign(nl, std::vector<bool>(nl, false));
    for (int l = 0; l < nl; ++l) {
        const int pj = m.link_parent_joint[l];
        if (pj >= 0) {
            const int pl = m.joint_parent_link[pj];
            m.acm_allowed[l][pl] = m.acm_allowed[pl][l] = true;
        }
    }
    for (const auto& pr : d.acm_allowed) {
        const int a = m.linkIndex(pr.first), b = m.linkIndex(pr.second);
        if (a < 0 || b < 0) return fail("acm references unknown link");
        m.acm_allowed[a][b] = m.acm_allowed[b][a] = true;
    }
    return true;
}

// joint transform functions (sbpl_collision_checking/src/transform_functions.h:95-258)
inline Affine joint_transform(const RobotCollisionModel& m, int jidx, double q)
{
    const Affine& o = m.joint_origins[jidx];
    Affine t;
    switch (m.joint_kinds[jidx]) {
    case TK_FIXED:
        return o;  // :251-258
    case TK_REV_X: {  // :104-137
        double s, c;
        det_sincos(q, &s, &c);
        for (int i = 0; i < 3; ++i) {
            t.m[i][0] = o.m[i][0];
            t.m[i][1] = c * o.m[i][1] + s * o.m[i][2];
            t.m[i][2] = c * o.m[i][2] - s * o.m[i][1];
            t.m[i][3] = o.m[i][3];
        }
        return t;
    }
    case TK_REV_Y: {  // :139-172
        double s, c;
        det_sincos(q, &s, &c);
        for (int i = 0; i < 3; ++i) {
            t.m[i][0] = c * o.m[i][0] - s * o.m[i][2];
            t.m[i][1] = o.m[i][1];
            t.m[i][2] = s * o.m[i][0] + c * o.m[i][2];
            t.m[i][3] = o.m[i][3];
        }
        return t;
    }
    case TK_REV_Z: {  // :174-207
        double s, c;
        det_sincos(q, &s, &c);
        for (int i = 0; i < 3; ++i) {
            t.m[i][0] = o.m[i][0] * c + o.m[i][1] * s;
            t.m[i][1] = o.m[i][1] * c - o.m[i][0] * s;
            t.m[i][2] = o.m[i][2];
            t.m[i][3] = o.m[i][3];
        }
        return t;
    }
    case TK_REV_GENERIC: {  // :96-102  o * AngleAxisd(q, axis)
        double R[3][3];
        angle_axis_matrix(q, m.joint_axes[jidx], R);
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) {
                t.m[i][j] = (o.m[i][0] * R[0][j] + o.m[i][1] * R[1][j]) + o.m[i][2] * R[2][j];
            }
            t.m[i][3] = o.m[i][3];
        }
        return t;
    }
    case TK_PRISMATIC: {  // :218-226 translates along local Z regardless of axis
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) t.m[i][j] = o.m[i][j];
            t.m[i][3] = ((o.m[i][0] * 0.0 + o.m[i][1] * 0.0) + o.m[i][2] * q) + o.m[i][3];
        }
        return t;
    }
    }
    return o;
}

// ---------------------------------------------------------------------------
// RobotCollisionState: link transforms and sphere positions
// (sbpl_collision_checking/include/sbpl_collision_checking/robot_collision_state.h:386-431, 561-581)
// The reference is lazy with dirty flags; values are pure functions of the
// joint variables so this computes per query with memoisation.
// ---------------------------------------------------------------------------

struct RobotCollisionState {
    const RobotCollisionModel* model = nullptr;
    std::vector<double> jvars;
    std::vector<Affine> link_tf;
    std::vector<char> link_ok;

    explicit RobotCollisionState(const RobotCollisionModel* m) : model(m)
    {
        jvars.assign(m->var_names.size(), 0.0);
        link_tf.assign(m->link_names.size(), Affine::Identity());
        link_ok.assign(m->link_names.size(), 0);
    }

    void setJointVarPositions(const double* v)
    {
        std::copy(v, v + jvars.size(), jvars.begin());
        std::fill(link_ok.begin(), link_ok.end(), 0);
        link_ok[0] = 1;  // root link = world->model transform = identity (call_planner.cpp:1658)
        link_tf[0] = Affine::Identity();
    }

    const Affine& linkTransform(int lidx)
    {
        if (link_ok[lidx]) return link_tf[lidx];
        const int pj = model->link_parent_joint[lidx];
        const int pl = model->joint_parent_link[pj];
        const int v = model->joint_var[pj];
        const Affine J = joint_transform(*model, pj, v >= 0 ? jvars[v] : 0.0);
        if (pl == 0) {
            // identity * J == J exactly (1*x + 0*y + 0*z); skip the product
            link_tf[lidx] = J;
        } else {
            link_tf[lidx] = mul(linkTransform(pl), J);
        }
        link_ok[lidx] = 1;
        return link_tf[lidx];
    }

    Vec3 spherePos(int smidx, int node)
    {
        const int l = model->spheres_model_link[smidx];
        return apply(linkTransform(l), model->spheres_models[smidx].nodes[node].center);
    }
};

// ---------------------------------------------------------------------------
// RobotMotionCollisionModel (robot_motion_collision_model.cpp:41-275, 371-407)
// ---------------------------------------------------------------------------

struct RobotMotionCollisionModel {
    std::vector<Vec3> mr_centers;
    std::vector<double> mr_radii;
    std::vector<double> k;  // per joint: |mr_center| + mr_radius

    explicit RobotMotionCollisionModel(const RobotCollisionModel& rcm)
    {
        const int nj = (int)rcm.joint_names.size();
        std::vector<int> q_joint(nj + 1, -1);
        int q_head = 0, q_tail = 0;
        std::vector<int> p_joint(nj, 0);
        for (size_t l = 0; l < rcm.link_names.size(); ++l) {
            if (rcm.link_child_joints[l].empty()) {
                const int j = rcm.link_parent_joint[l];
                if (j >= 0) q_joint[q_tail++] = j;
            }
        }
        mr_centers.assign(nj, Vec3());
        mr_radii.assign(nj, 0.0);
        std::vector<std::vector<Vec3>> sample_spheres(nj);
        std::vector<double> sample_radii(nj, 0.0);
        while (q_head != q_tail) {
            const int jidx = q_joint[q_head++];
            std::vector<Vec3> centers;
            std::vector<double> radii;
            const int cl = rcm.joint_child_link[jidx];
            if (rcm.link_spheres_model[cl] >= 0) {
                const SphereTree& t = rcm.spheres_models[rcm.link_spheres_model[cl]];
                centers.push_back(t.nodes[t.root()].center);
                radii.push_back(t.nodes[t.root()].radius);
            }
            for (int cj : rcm.link_child_joints[cl]) {
                if (sample_radii[cj] != 0.0) {
                    const Affine& T = rcm.joint_origins[cj];
                    for (const Vec3& p : sample_spheres[cj]) {
                        centers.push_back(apply(T, p));
                        radii.push_back(sample_radii[cj]);
                    }
                }
            }
            Vec3 mr_center;
            double mr_radius = 0.0;
            if (!centers.empty()) {
                for (const Vec3& c : centers) mr_center = mr_center + c;
                const double n = (double)centers.size();
                mr_center = {mr_center.x / n, mr_center.y / n, mr_center.z / n};
                for (size_t i = 0; i < centers.size(); ++i) {
                    const double radius = norm(centers[i] - mr_center) + radii[i];
                    mr_radius = std::max(mr_radius, radius);
                }
            }
            mr_centers[jidx] = mr_center;
            mr_radii[jidx] = mr_radius;

            std::vector<Vec3> samples;
            if (mr_radius != 0.0) {
                double res = 2.0 * M_PI / 180.0;
                auto rot_sample = [&](double val) {
                    double R[3][3];
                    angle_axis_matrix(val, rcm.joint_axes[jidx], R);
                    Vec3 p;
                    p.x = (R[0][0] * mr_center.x + R[0][1] * mr_center.y) + R[0][2] * mr_center.z;
                    p.y = (R[1][0] * mr_center.x + R[1][1] * mr_center.y) + R[1][2] * mr_center.z;
                    p.z = (R[2][0] * mr_center.x + R[2][1] * mr_center.y) + R[2][2] * mr_center.z;
                    samples.push_back(p);
                };
                if (rcm.joint_types[jidx] == REVOLUTE) {
                    const int v = rcm.joint_var[jidx];
                    const double span = rcm.var_max[v] - rcm.var_min[v];
                    const int sample_count = (int)std::round(span / res) + 1;
                    for (int i = 0; i < sample_count; ++i) {
                        const double alpha = (double)i / (double)(sample_count - 1);
                        const double val = (1.0 - alpha) * rcm.var_min[v] + alpha * rcm.var_max[v];
                        rot_sample(val);
                    }
                } else if (rcm.joint_types[jidx] == CONTINUOUS) {
                    const int sample_count = (int)std::round(2.0 * M_PI / res);
                    res = 2.0 * M_PI / sample_count;
                    for (int i = 0; i < sample_count; ++i) rot_sample(i * res);
                } else if (rcm.joint_types[jidx] == PRISMATIC) {
                    const int v = rcm.joint_var[jidx];
                    const double span = rcm.var_max[v] - rcm.var_min[v];
                    const int sample_count = (int)std::round(span / res) + 1;
                    for (int i = 0; i < sample_count; ++i) {
                        const double alpha = (double)i / (double)(sample_count - 1);
                        const double val = (1.0 - alpha) * rcm.var_min[v] + alpha * rcm.var_max[v];
                        samples.push_back(mr_center + val * rcm.joint_axes[jidx]);
                    }
                } else {  // FIXED :213-217
                    samples.push_back(apply(rcm.joint_origins[jidx], mr_center));
                }
            }
            sample_spheres[jidx] = samples;
            sample_radii[jidx] = mr_radius;

            const int pl = rcm.joint_parent_link[jidx];
            if (pl >= 0) {
                const int pj = rcm.link_parent_joint[pl];
                if (pj >= 0) {
                    ++p_joint[pj];
                    if (p_joint[pj] == (int)rcm.link_child_joints[pl].size()) q_joint[q_tail++] = pj;
                }
            }
        }
        k.assign(nj, 0.0);
        for (int j = 0; j < nj; ++j) k[j] = norm(mr_centers[j]) + mr_radii[j];
    }
};

// ---------------------------------------------------------------------------
// OccupancyGrid / DistanceMap lookup side
// (smpl/include/smpl/distance_map/detail/distance_map.hpp:112-150 ctor, 281-300 lookup,
//  520-536 worldToGrid/isCellValid; smpl/include/smpl/occupancy_grid.h:221-237)
// Construction (propagation) is SURVEY row N1 ("next"); the squared cell
// distances are an input here, laid out like the reference's Grid3: x-major, z fastest.
// ---------------------------------------------------------------------------

struct OccupancyGrid {
    double origin[3] = {0, 0, 0};
    double res = 0.02, inv_res = 50.0, max_dist = 0.4;
    int n[3] = {0, 0, 0};
    int dmax_int = 0, dmax_sqrd_int = 0;
    std::vector<int> d2;             // interior cells only
    std::vector<double> sqrt_table;  // res * sqrt(i)
    mutable long lookups = 0;        // counts getSquaredDist calls (roofline accounting)
    // The reference's own storage, for timing only ("dense grid with the reference's 48 B AoS cell and a
    // compact 4 B variant, both stated"): DistanceMap keeps an array-of-structures of 48-byte cells over the PADDED grid
    // (distance_map.h:110-127: x, y, z, dist, dist_new, counter, obs*, bucket, dir, pos), so a lookup touches one 48-byte
    // record of an 824 MB array at 256^3 instead of 4 bytes of a 67 MB one.  Same values, same results.
    struct RefCell { int x, y, z, dist, dist_new, counter; void* obs; int bucket, dir, pos; };
    static_assert(sizeof(RefCell) == 48, "the reference's cell is 48 bytes");
    std::vector<RefCell> aos;
    bool use_aos = false;
    void buildAosCells()
    {
        const size_t px = (size_t)n[0] + 2, py = (size_t)n[1] + 2, pz = (size_t)n[2] + 2;
        aos.assign(px * py * pz, RefCell{0, 0, 0, 0, 0, 0, nullptr, -1, 0, 0});
        for (int x = 0; x < n[0]; ++x)
            for (int y = 0; y < n[1]; ++y)
                for (int z = 0; z < n[2]; ++z) {
                    RefCell& c = aos[((size_t)(x + 1) * py + (y + 1)) * pz + (z + 1)];
                    c.x = x + 1; c.y = y + 1; c.z = z + 1;
                    c.dist = c.dist_new = d2[((size_t)x * n[1] + y) * n[2] + z];
                }
        use_aos = true;
    }

    void init(const double o[3], int nx, int ny, int nz, double resolution, double maxd, const int* cells)
    {
        origin[0] = o[0]; origin[1] = o[1]; origin[2] = o[2];
        res = resolution;
        inv_res = 1.0 / resolution;
        max_dist = maxd;
        n[0] = nx; n[1] = ny; n[2] = nz;
        dmax_int = (int)std::ceil(max_dist * inv_res);
        dmax_sqrd_int = dmax_int * dmax_int;
        d2.assign(cells, cells + (size_t)nx * ny * nz);
        sqrt_table.resize(dmax_sqrd_int + 1);
        for (int i = 0; i < dmax_sqrd_int + 1; ++i) sqrt_table[i] = res * std::sqrt((double)i);
    }
    void worldToGrid(double wx, double wy, double wz, int& x, int& y, int& z) const
    {
        x = (int)(inv_res * (wx - (origin[0] - res)) + 0.5) - 1;
        y = (int)(inv_res * (wy - (origin[1] - res)) + 0.5) - 1;
        z = (int)(inv_res * (wz - (origin[2] - res)) + 0.5) - 1;
    }
    bool isCellValid(int x, int y, int z) const
    {
        return x >= 0 && x < n[0] && y >= 0 && y < n[1] && z >= 0 && z < n[2];
    }
    double getCellDistance(int x, int y, int z) const
    {
        if (!isCellValid(x, y, z)) return 0.0;
        if (use_aos) return sqrt_table[aos[((size_t)(x + 1) * ((size_t)n[1] + 2) + (y + 1)) * ((size_t)n[2] + 2) + (z + 1)].dist];
        return sqrt_table[d2[((size_t)x * n[1] + y) * n[2] + z]];
    }
    double getMetricDistance(double x, double y, double z) const
    {
        int gx, gy, gz;
        worldToGrid(x, y, z, gx, gy, gz);
        return getCellDistance(gx, gy, gz);
    }
    // distance_map_interface.h:113-114
    double getSquaredDist(double x, double y, double z) const
    {
        ++lookups;
        const double d = getMetricDistance(x, y, z);
        return d * d;
    }
};

// ---------------------------------------------------------------------------
// collision checking (collision_operations.h:67-164; self_collision_model.cpp:407-428,
// 839-899, 1093-1268; collision_space.cpp:532-581)
// ---------------------------------------------------------------------------

enum TraversalOrder {
    ORDER_REFERENCE = 0,  // one LIFO stack seeded with all group roots (self_collision_model.cpp:839-861)
    ORDER_CHAIN = 1       // trees one by one in group order (what the HIP kernel does; same booleans)
};

struct CollisionSpace {
    const OccupancyGrid* grid = nullptr;
    const RobotCollisionModel* rcm = nullptr;
    RobotMotionCollisionModel rmcm;
    RobotCollisionState rcs;
    std::vector<int> planning_to_var;  // planning joint i -> model var index
    std::vector<double> joint_vars;    // full model variable vector
    std::vector<std::pair<int, int>> checked_pairs;  // spheres-model index pairs
    std::vector<int> chain_order;                    // group trees in depth-first link order
    double padding = 0.0;
    TraversalOrder order = ORDER_REFERENCE;

    CollisionSpace(const OccupancyGrid* g, const RobotCollisionModel* m, const std::vector<std::string>& planning_joints)
        : grid(g), rcm(m), rmcm(*m), rcs(m)
    {
        for (const std::string& j : planning_joints) planning_to_var.push_back(m->varIndex(j));
        joint_vars.assign(m->var_names.size(), 0.0);
        // ORDER_CHAIN: group trees in depth-first link order (children in file order)
        {
            std::vector<int> stack{0};
            while (!stack.empty()) {
                const int l = stack.back();
                stack.pop_back();
                const int sm = m->link_spheres_model[l];
                if (sm >= 0 && std::find(m->group_spheres_models.begin(), m->group_spheres_models.end(), sm) != m->group_spheres_models.end())
                    chain_order.push_back(sm);
                const auto& ch = m->link_child_joints[l];
                for (auto it = ch.rbegin(); it != ch.rend(); ++it) stack.push_back(m->joint_child_link[*it]);
            }
        }
        // self_collision_model.cpp:1233-1268 updateRobotCheckedSphereIndices
        const auto& gl = m->group_links;
        for (size_t l1 = 0; l1 < gl.size(); ++l1) {
            if (m->link_spheres_model[gl[l1]] < 0) continue;
            for (size_t l2 = l1 + 1; l2 < gl.size(); ++l2) {
                if (m->link_spheres_model[gl[l2]] < 0) continue;
                if (!m->acm_allowed[gl[l1]][gl[l2]]) {
                    checked_pairs.emplace_back(m->link_spheres_model[gl[l1]], m->link_spheres_model[gl[l2]]);
                }
            }
        }
    }

    // collision_operations.h:67-77
    bool checkSphere(Vec3 p, double radius) const
    {
        const double er = radius + padding;
        const double d = grid->getSquaredDist(p.x, p.y, p.z);
        return d >= er * er;
    }

    // collision_operations.h:105-164 with the queue discipline of the caller
    bool checkVoxels()
    {
        struct Item { int sm, node; };
        std::vector<Item> q;
        auto run = [&]() {
            while (!q.empty()) {
                const Item it = q.back();
                q.pop_back();
                const SphereTree& t = rcm->spheres_models[it.sm];
                const SphereNode& s = t.nodes[it.node];
                if (checkSphere(rcs.spherePos(it.sm, it.node), s.radius)) continue;
                if (s.isLeaf()) return false;
                const SphereNode& l = t.nodes[s.left];
                const SphereNode& r = t.nodes[s.right];
                if (l.radius > r.radius) { q.push_back({it.sm, s.right}); q.push_back({it.sm, s.left}); }
                else { q.push_back({it.sm, s.left}); q.push_back({it.sm, s.right}); }
            }
            return true;
        };
        if (order == ORDER_REFERENCE) {
            for (int sm : rcm->group_spheres_models) q.push_back({sm, rcm->spheres_models[sm].root()});
            return run();
        }
        for (int sm : chain_order) {
            q.clear();
            q.push_back({sm, rcm->spheres_models[sm].root()});
            if (!run()) return false;
        }
        return true;
    }

    // self_collision_model.cpp:1093-1218
    bool checkSpheresPair(int sm1, int sm2)
    {
        const SphereTree& t1 = rcm->spheres_models[sm1];
        const SphereTree& t2 = rcm->spheres_models[sm2];
        std::vector<std::pair<int, int>> q;
        q.emplace_back(t1.root(), t2.root());
        while (!q.empty()) {
            const auto pr = q.back();
            q.pop_back();
            const SphereNode& s1 = t1.nodes[pr.first];
            const SphereNode& s2 = t2.nodes[pr.second];
            const Vec3 p1 = rcs.spherePos(sm1, pr.first);
            const Vec3 p2 = rcs.spherePos(sm2, pr.second);
            const double cd2 = sqnorm(p2 - p1);
            const double rr = s1.radius + s2.radius;
            const double cr2 = rr * rr;
            if (cd2 > cr2) continue;
            if (s1.isLeaf() && s2.isLeaf()) {
                // :1136 ACM lookup is by *sphere* name, which never matches a
                // link entry -> always a collision (SURVEY a18 parity note)
                return false;
            }
            bool split1;
            if (s1.isLeaf()) split1 = false;
            else if (s2.isLeaf()) split1 = true;
            else split1 = s1.radius > s2.radius;
            if (split1) {
                const Vec3 pl = rcs.spherePos(sm1, s1.left), prr = rcs.spherePos(sm1, s1.right);
                const double cl = sqnorm(p2 - pl), cr = sqnorm(p2 - prr);
                if (cl < cr) { q.emplace_back(s1.right, pr.second); q.emplace_back(s1.left, pr.second); }
                else { q.emplace_back(s1.left, pr.second); q.emplace_back(s1.right, pr.second); }
            } else {
                const Vec3 pl = rcs.spherePos(sm2, s2.left), prr = rcs.spherePos(sm2, s2.right);
                const double cl = sqnorm(p1 - pl), cr = sqnorm(p1 - prr);
                if (cl < cr) { q.emplace_back(pr.first, s2.right); q.emplace_back(pr.first, s2.left); }
                else { q.emplace_back(pr.first, s2.left); q.emplace_back(pr.first, s2.right); }
            }
        }
        return true;
    }

    // collision_space.cpp:741-774 + self_collision_model.cpp:407-428
    bool isStateValid(const std::vector<double>& state)
    {
        for (size_t i = 0; i < planning_to_var.size(); ++i) joint_vars[planning_to_var[i]] = state[i];
        rcs.setJointVarPositions(joint_vars.data());
        if (!checkVoxels()) return false;
        for (const auto& pr : checked_pairs) {
            if (!checkSpheresPair(pr.first, pr.second)) return false;
        }
        return true;
    }

    // robot_motion_collision_model.cpp:371-407 (variables subset form)
    double maxSphereMotion(const std::vector<double>& start, const std::vector<double>& finish) const
    {
        double motion = 0.0;
        for (size_t i = 0; i < start.size(); ++i) {
            const int jidx = rcm->var_joint[planning_to_var[i]];
            double dist;
            switch (rcm->joint_types[jidx]) {
            case CONTINUOUS:
                dist = shortest_angle_dist(finish[i], start[i]);
                motion += rmcm.k[jidx] * dist;
                break;
            case REVOLUTE:
                dist = std::fabs(finish[i] - start[i]);
                motion += rmcm.k[jidx] * dist;
                break;
            case PRISMATIC:
                dist = std::fabs(finish[i] - start[i]);
                motion += dist;
                break;
            default:
                break;
            }
        }
        return motion;
    }

    // robot_motion_collision_model.h:173-181, 352-366
    int waypointCount(const std::vector<double>& start, const std::vector<double>& finish) const
    {
        const double max_motion = maxSphereMotion(start, finish);
        if (max_motion == 0.0) return 0;
        const int wc = (int)std::ceil(max_motion / 0.05) + 1;
        return std::max(2, wc);
    }

    // robot_motion_collision_model.h:221-247 (diffs), 297-320 (interpolate)
    void interpolationDiffs(const std::vector<double>& start, const std::vector<double>& finish, std::vector<double>& diffs) const
    {
        diffs.resize(start.size());
        for (size_t i = 0; i < start.size(); ++i) {
            const int jidx = rcm->var_joint[planning_to_var[i]];
            if (rcm->joint_types[jidx] == CONTINUOUS) diffs[i] = shortest_angle_diff(finish[i], start[i]);
            else diffs[i] = finish[i] - start[i];
        }
    }

    // collision_space.cpp:538-581
    bool isStateToStateValid(const std::vector<double>& start, const std::vector<double>& finish)
    {
        const int W = waypointCount(start, finish);
        std::vector<double> diffs;
        interpolationDiffs(start, finish, diffs);
        const double inv = (W > 0) ? 1.0 / (double)(W - 1) : 0.0;
        std::vector<double> interm(start.size());
        auto check = [&](int n) {
            const double alpha = (double)n * inv;
            for (size_t i = 0; i < start.size(); ++i) interm[i] = start[i] + alpha * diffs[i];
            return isStateValid(interm);
        };
        const int inc_cc = 5;
        if (W > inc_cc) {
            for (int i = 0; i < inc_cc; ++i) {
                for (int j = i; j < W; j += inc_cc) {
                    if (!check(j)) return false;
                }
            }
        } else {
            for (int i = 0; i < W; ++i) {
                if (!check(i)) return false;
            }
        }
        return true;
    }
};

// ---------------------------------------------------------------------------
// planning robot model: FK of the planning link + joint limits
// (sbpl_kdl_robot_model/src/kdl_robot_model.cpp:173-235, 326-337, 400-423).
// orocos-kdl is not in the tree: FK arithmetic is the serial-chain product of
// the arithmetic contract (same as the collision FK), parity unpinned.
// ---------------------------------------------------------------------------

struct PlanningRobotModel {
    const RobotCollisionModel* rcm = nullptr;
    std::vector<int> planning_to_var;
    std::vector<int> chain;  // joints from root to planning link
    std::vector<double> min_limits, max_limits;
    std::vector<bool> continuous, bounded;

    bool init(const RobotCollisionModel* m, const std::vector<std::string>& planning_joints, const std::string& planning_link)
    {
        rcm = m;
        for (const std::string& j : planning_joints) {
            const int v = m->varIndex(j);
            if (v < 0) return false;
            planning_to_var.push_back(v);
            const bool c = m->var_continuous[v];
            continuous.push_back(c);
            bounded.push_back(!c);
            // kdl_robot_model.cpp:299-303: continuous joints report [-pi, pi]
            min_limits.push_back(c ? -M_PI : m->var_min[v]);
            max_limits.push_back(c ? M_PI : m->var_max[v]);
        }
        int l = m->linkIndex(planning_link);
        if (l < 0) return false;
        while (m->link_parent_joint[l] >= 0) {
            const int j = m->link_parent_joint[l];
            chain.push_back(j);
            l = m->joint_parent_link[j];
        }
        std::reverse(chain.begin(), chain.end());
        return true;
    }
    int jointVariableCount() const { return (int)planning_to_var.size(); }

    // kdl_robot_model.cpp:173-189
    static double normalizeAngle(double a, double a_min, double a_max)
    {
        if (std::fabs(a) > 2.0 * M_PI) a = std::fmod(a, 2.0 * M_PI);
        while (a > a_max) a -= 2.0 * M_PI;
        while (a < a_min) a += 2.0 * M_PI;
        return a;
    }
    // kdl_robot_model.cpp:210-235, 326-337
    bool checkJointLimits(const std::vector<double>& angles) const
    {
        for (size_t i = 0; i < angles.size(); ++i) {
            if (min_limits[i] > max_limits[i]) return false;
        }
        for (size_t i = 0; i < angles.size(); ++i) {
            const double min_angle_norm = normalize_angle(min_limits[i]);
            const double a = normalizeAngle(angles[i], min_limits[i], min_angle_norm);
            if (a < min_limits[i] || a > max_limits[i]) return false;
        }
        return true;
    }
    // kdl_robot_model.cpp:400-423 (position part; rpy is not used by the BFS heuristic)
    Affine computePlanningLinkFrame(const std::vector<double>& angles) const
    {
        std::vector<double> full(rcm->var_names.size(), 0.0);
        for (size_t i = 0; i < angles.size(); ++i) {
            // :191-198 normalizeAngles before FK
            full[planning_to_var[i]] = continuous[i] ? normalize_angle(angles[i]) : angles[i];
        }
        Affine T = Affine::Identity();
        bool first = true;
        for (int j : chain) {
            const int v = rcm->joint_var[j];
            const Affine J = joint_transform(*rcm, j, v >= 0 ? full[v] : 0.0);
            if (first) { T = J; first = false; }
            else T = mul(T, J);
        }
        return T;
    }
    void computePlanningLinkFK(const std::vector<double>& angles, double xyz[3]) const
    {
        const Affine T = computePlanningLinkFrame(angles);
        xyz[0] = T.m[0][3]; xyz[1] = T.m[1][3]; xyz[2] = T.m[2][3];
    }
};

// ---------------------------------------------------------------------------
// BFS_3D (smpl/src/bfs3d.cpp:40-111, 132-141, 156-201, 507-547; bfs3d.h:151-220)
// The background thread + busy-wait of the reference is replaced by running to
// completion (identical distances; SURVEY section 7 "BFS thread semantics").
// ---------------------------------------------------------------------------

struct BFS_3D {
    static const int WALL = 0x7FFFFFFF;
    static const int UNDISCOVERED = -1;
    int dim_x = 0, dim_y = 0, dim_z = 0, dim_xy = 0, dim_xyz = 0;
    std::vector<int> dist;
    std::vector<int> queue;

    BFS_3D(int width, int height, int length)
    {
        dim_x = width + 2; dim_y = height + 2; dim_z = length + 2;
        dim_xy = dim_x * dim_y;
        dim_xyz = dim_xy * dim_z;
        dist.resize(dim_xyz);
        queue.resize((size_t)width * height * length);
        for (int node = 0; node < dim_xyz; ++node) {
            const int x = node % dim_x, y = node / dim_x % dim_y, z = node / dim_xy;
            if (x == 0 || x == dim_x - 1 || y == 0 || y == dim_y - 1 || z == 0 || z == dim_z - 1) dist[node] = WALL;
            else dist[node] = UNDISCOVERED;
        }
    }
    bool inBounds(int x, int y, int z) const
    {
        return !(x < 0 || y < 0 || z < 0 || x >= dim_x - 2 || y >= dim_y - 2 || z >= dim_z - 2);
    }
    int getNode(int x, int y, int z) const
    {
        if (!inBounds(x, y, z)) return -1;
        return (z + 1) * dim_xy + (y + 1) * dim_x + (x + 1);
    }
    void setWall(int x, int y, int z) { dist[getNode(x, y, z)] = WALL; }
    bool isWall(int x, int y, int z) const { return dist[getNode(x, y, z)] == WALL; }
    int getDistance(int x, int y, int z) const { return dist[getNode(x, y, z)]; }

    int run(int x, int y, int z)
    {
        for (int i = 0; i < dim_xyz; ++i) {
            if (dist[i] != WALL) dist[i] = UNDISCOVERED;
        }
        const int origin = getNode(x, y, z);
        if (origin == -1) return 0;
        int head = 0, tail = 1;
        queue[0] = origin;
        dist[origin] = 0;  // note: overwrites a WALL at the goal cell, as the reference does
        const int w = dim_x, p = dim_xy;
        const int offs[26] = {-w, 1, w, -1, -w - 1, -w + 1, w + 1, w - 1, p, -w + p, 1 + p, w + p, -1 + p,
                              -w - 1 + p, -w + 1 + p, w + 1 + p, w - 1 + p, -p, -w - p, 1 - p, w - p, -1 - p,
                              -w - 1 - p, -w + 1 - p, w + 1 - p, w - 1 - p};
        while (head < tail) {
            const int cur = queue[head++];
            const int cost = dist[cur] + 1;
            for (int o : offs) {
                if (dist[cur + o] < 0) {
                    queue[tail++] = cur + o;
                    dist[cur + o] = cost;
                }
            }
        }
        return 1;
    }
};

// ---------------------------------------------------------------------------
// goal + BfsHeuristic (smpl/src/heuristic/bfs_heuristic.cpp:52-163, 331-366;
// smpl/include/smpl/heuristic/robot_heuristic.h:62 Infinity = INT16_MAX)
// ---------------------------------------------------------------------------

enum GoalType { XYZ_GOAL = 0, XYZ_RPY_GOAL = 1, JOINT_STATE_GOAL = 2 };

struct GoalConstraint {
    GoalType type = JOINT_STATE_GOAL;
    std::vector<double> angles, angle_tolerances;
    double tgt_off_pose[6] = {0, 0, 0, 0, 0, 0};
    double xyz_tolerance[3] = {0, 0, 0};
};

struct BfsHeuristic {
    const OccupancyGrid* grid = nullptr;
    std::unique_ptr<BFS_3D> bfs;
    double inflation_radius = 0.0;
    int cost_per_cell = 1;
    static const int Infinity = 32767;

    void init(const OccupancyGrid* g, double radius, int cpc)
    {
        grid = g;
        inflation_radius = radius;
        cost_per_cell = cpc;
        syncGridAndBfs();
    }
    // bfs_heuristic.cpp:331-353
    void syncGridAndBfs()
    {
        const int xc = grid->n[0], yc = grid->n[1], zc = grid->n[2];
        bfs.reset(new BFS_3D(xc, yc, zc));
        for (int x = 0; x < xc; ++x)
            for (int y = 0; y < yc; ++y)
                for (int z = 0; z < zc; ++z)
                    if (grid->getCellDistance(x, y, z) <= inflation_radius) bfs->setWall(x, y, z);
    }
    // bfs_heuristic.cpp:83-101
    void updateGoal(const GoalConstraint& goal)
    {
        int gx, gy, gz;
        grid->worldToGrid(goal.tgt_off_pose[0], goal.tgt_off_pose[1], goal.tgt_off_pose[2], gx, gy, gz);
        bfs->run(gx, gy, gz);
    }
    // bfs_heuristic.cpp:129-138
    double getMetricGoalDistance(double x, double y, double z) const
    {
        int gx, gy, gz;
        grid->worldToGrid(x, y, z, gx, gy, gz);
        if (!bfs->inBounds(gx, gy, gz)) return (double)BFS_3D::WALL * grid->res;
        return (double)bfs->getDistance(gx, gy, gz) * grid->res;
    }
    // bfs_heuristic.cpp:355-366
    int getBfsCostToGoal(int x, int y, int z) const
    {
        if (!bfs->inBounds(x, y, z)) return Infinity;
        if (bfs->getDistance(x, y, z) == BFS_3D::WALL) return Infinity;
        return cost_per_cell * bfs->getDistance(x, y, z);
    }
    // bfs_heuristic.cpp:148-163 with the point already projected
    int heuristicAtPoint(const double p[3]) const
    {
        int x, y, z;
        grid->worldToGrid(p[0], p[1], p[2], x, y, z);
        return getBfsCostToGoal(x, y, z);
    }
};

// ---------------------------------------------------------------------------
// motion primitives + action space
// (smpl/src/graph/manip_lattice_action_space.cpp:49-261, 376-449, 507-621, 662-691)
// ---------------------------------------------------------------------------

enum MprimType { LONG_DISTANCE = -1, SNAP_TO_RPY = 0, SNAP_TO_XYZ = 1, SNAP_TO_XYZ_RPY = 2, SHORT_DISTANCE = 3, NUM_MPRIM_TYPES = 4 };

struct MotionPrimitive {
    MprimType type = LONG_DISTANCE;
    std::vector<double> delta;  // one waypoint (the loader only ever makes one)
    int group = -1;
    double weight = 1.0;
};

struct ActionSpaceParams {
    bool use_long_and_short = false;
    bool enabled[NUM_MPRIM_TYPES] = {false, false, false, false};  // snaps + short; LONG is always enabled
    double thresh[NUM_MPRIM_TYPES] = {0.4, 0.4, 0.4, 0.4};
    bool xy_rotate_by_var3 = true;  // [FORK] manip_lattice_action_space.cpp:590-599
};

struct ActionSpace {
    std::vector<MotionPrimitive> mprims;
    ActionSpaceParams params;

    // manip_lattice_action_space.cpp:233-261
    void clear()
    {
        mprims.clear();
        MotionPrimitive m;
        m.weight = 0.5;
        m.group = -1;
        m.type = SNAP_TO_RPY; mprims.push_back(m);
        m.type = SNAP_TO_XYZ; mprims.push_back(m);
        m.type = SNAP_TO_XYZ_RPY; mprims.push_back(m);
    }
    // :202-228
    void addMotionPrim(const std::vector<double>& d, int group, double weight, bool short_dist, bool add_converse = true)
    {
        MotionPrimitive m;
        m.type = short_dist ? SHORT_DISTANCE : LONG_DISTANCE;
        m.delta = d;
        m.group = group;
        m.weight = weight;
        mprims.push_back(m);
        if (add_converse) {
            for (double& v : m.delta) v *= -1.0;
            mprims.push_back(m);
        }
    }
    // :103-195.  Accepts both layouts: cols == nvars -> upstream rows
    // (group ANY, weight 1.0); cols == nvars + 2 -> fork rows (deltas, group, weight).
    bool load(const std::string& text, const std::vector<double>& resolutions, std::string* err = nullptr)
    {
        std::istringstream in(text);
        std::string hdr;
        int nrows = 0, ncols = 0, nshort = 0;
        if (!(in >> hdr) || hdr != "Motion_Primitives(degrees):") { if (err) *err = "bad header"; return false; }
        if (!(in >> nrows >> ncols >> nshort)) { if (err) *err = "bad counts"; return false; }
        const int nv = (int)resolutions.size();
        const bool fork_fmt = (ncols == nv + 2);
        if (!fork_fmt && ncols != nv) { if (err) *err = "column count does not match the robot"; return false; }
        for (int i = 0; i < nrows; ++i) {
            std::vector<double> d(nv);
            for (int j = 0; j < nv; ++j) {
                double v;
                if (!(in >> v)) { if (err) *err = "short row"; return false; }
                d[j] = v * resolutions[j];  // :172 delta in multiples of the variable resolution
            }
            int group = -1;
            double weight = 1.0;
            if (fork_fmt && !(in >> group >> weight)) { if (err) *err = "missing group/weight"; return false; }
            addMotionPrim(d, group, weight, !(i < nrows - nshort));
        }
        return true;
    }
    // :662-691
    bool mprimActive(double goal_dist, MprimType type) const
    {
        if (type == LONG_DISTANCE) {
            if (params.use_long_and_short) return true;
            const bool near_goal = goal_dist <= params.thresh[SHORT_DISTANCE];
            return !(params.enabled[SHORT_DISTANCE] && near_goal);
        } else if (type == SHORT_DISTANCE) {
            if (params.use_long_and_short) return params.enabled[type];
            const bool near_goal = goal_dist <= params.thresh[type];
            return params.enabled[type] && near_goal;
        }
        return params.enabled[type] && goal_dist <= params.thresh[type];
    }
    // :575-621 (successor direction only)
    bool applyMotionPrimitive(const std::vector<double>& state, const MotionPrimitive& mp, std::vector<double>& out) const
    {
        if (mp.delta.size() != state.size()) return false;
        out = mp.delta;
        if (params.xy_rotate_by_var3 && state.size() > 3) {
            double s, c;
            det_sincos(state[3], &s, &c);
            // Eigen 2x2 * 2-vector: row0 = c*a0 + (-s)*a1 ; row1 = s*a0 + c*a1
            const double a0 = mp.delta[0], a1 = mp.delta[1];
            out[0] = c * a0 + (-s) * a1;
            out[1] = s * a0 + c * a1;
        }
        for (size_t j = 0; j < state.size(); ++j) out[j] = out[j] + state[j];
        return true;
    }
};

// ---------------------------------------------------------------------------
// ManipLattice (smpl/src/graph/manip_lattice.cpp:72-149 init, 219-313 GetSuccs,
// 1245-1354 coords + state table, 1414-1437 cost, 1511-1580 checkAction,
// 1582-1696 isGoal, 1944-1981 setStart)
// ---------------------------------------------------------------------------

struct CoordHash {
    size_t operator()(const std::vector<int>& c) const
    {
        // the id of a state depends only on insertion order, not on the hash
        size_t seed = 0;
        for (int v : c) seed ^= std::hash<int>()(v) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
        return seed;
    }
};

struct SuccessorRecord {  // one evaluated (state, primitive) pair; what the HIP kernel is compared against
    int prim = -1;
    int flags = 0;        // 1 valid, 2 goal, 0x10 inactive, 0x20 joint limits, 0x40 collision
    std::vector<double> q;
    std::vector<int> coord;
    int h = 0;
    int cost = 0;
    long lookups = 0;
};

struct ManipLattice {
    PlanningRobotModel* robot = nullptr;
    CollisionSpace* checker = nullptr;
    BfsHeuristic* heur = nullptr;
    ActionSpace* actions = nullptr;
    std::vector<double> min_limits, max_limits;
    std::vector<bool> continuous, bounded;
    std::vector<int> coord_vals;
    std::vector<double> coord_deltas;
    std::vector<std::vector<int>> state_coords;
    std::vector<std::vector<double>> state_angles;
    std::unordered_map<std::vector<int>, int, CoordHash> state_to_id;
    int goal_state_id = -1, start_state_id = -1;
    GoalConstraint goal;
    long succ_evals = 0;      // (state, active primitive) pairs carried through the loop body
    long expansions = 0;
    std::vector<SuccessorRecord>* trace = nullptr;  // optional per-evaluation dump of the last GetSuccs

    bool init(PlanningRobotModel* r, CollisionSpace* c, BfsHeuristic* h, ActionSpace* a, const std::vector<double>& resolutions)
    {
        robot = r; checker = c; heur = h; actions = a;
        const int n = r->jointVariableCount();
        if ((int)resolutions.size() != n) return false;
        min_limits = r->min_limits; max_limits = r->max_limits;
        continuous = r->continuous; bounded = r->bounded;
        goal_state_id = reserveHashEntry();  // :122 id 0 = goal
        coord_vals.resize(n);
        coord_deltas.resize(n);
        for (int v = 0; v < n; ++v) {  // :125-139
            if (continuous[v]) {
                coord_vals[v] = (int)std::round((2.0 * M_PI) / resolutions[v]);
                coord_deltas[v] = (2.0 * M_PI) / (double)coord_vals[v];
            } else if (bounded[v]) {
                const double span = std::fabs(max_limits[v] - min_limits[v]);
                coord_vals[v] = std::max(1, (int)std::round(span / resolutions[v]));
                coord_deltas[v] = span / (double)coord_vals[v];
            } else {
                coord_vals[v] = std::numeric_limits<int>::max();
                coord_deltas[v] = resolutions[v];
            }
        }
        return true;
    }
    // :1263-1289
    void stateToCoord(const std::vector<double>& state, std::vector<int>& coord) const
    {
        coord.resize(state.size());
        for (size_t i = 0; i < state.size(); ++i) {
            if (continuous[i]) {
                const double pos_angle = normalize_angle_positive(state[i]);
                coord[i] = (int)((pos_angle + coord_deltas[i] * 0.5) / coord_deltas[i]);
                if (coord[i] == coord_vals[i]) coord[i] = 0;
            } else if (!bounded[i]) {
                if (state[i] >= 0.0) coord[i] = (int)(state[i] / coord_deltas[i] + 0.5);
                else coord[i] = (int)(state[i] / coord_deltas[i] - 0.5);
            } else {
                coord[i] = (int)(((state[i] - min_limits[i]) / coord_deltas[i]) + 0.5);
            }
        }
    }
    // :1245-1261
    void coordToState(const std::vector<int>& coord, std::vector<double>& state) const
    {
        state.resize(coord.size());
        for (size_t i = 0; i < coord.size(); ++i) {
            if (continuous[i]) state[i] = coord[i] * coord_deltas[i];
            else if (!bounded[i]) state[i] = (double)coord[i] * coord_deltas[i];
            else state[i] = min_limits[i] + coord[i] * coord_deltas[i];
        }
    }
    int reserveHashEntry()  // :1336-1354
    {
        state_coords.emplace_back();
        state_angles.emplace_back();
        return (int)state_coords.size() - 1;
    }
    int getOrCreateState(const std::vector<int>& coord, const std::vector<double>& state)  // :1302-1334
    {
        auto it = state_to_id.find(coord);
        if (it != state_to_id.end()) return it->second;
        const int id = reserveHashEntry();
        state_coords[id] = coord;
        state_angles[id] = state;
        state_to_id[coord] = id;
        return id;
    }
    // :1582-1696 (JOINT_STATE and XYZ goals; XYZ_RPY needs IK + Eigen quaternions -> out of scope)
    bool isGoal(const std::vector<double>& state) const
    {
        if (goal.type == JOINT_STATE_GOAL) {
            std::vector<int> gc, cc;
            stateToCoord(goal.angles, gc);
            stateToCoord(state, cc);
            for (size_t i = 0; i < goal.angles.size(); ++i) {
                // [FORK] cells compared against a tolerance in radians (:1596-1606)
                if (std::fabs((double)(cc[i] - gc[i])) > goal.angle_tolerances[i]) return false;
            }
            return true;
        }
        if (goal.type == XYZ_GOAL) {
            double p[3];
            robot->computePlanningLinkFK(state, p);
            return std::fabs(p[0] - goal.tgt_off_pose[0]) <= goal.xyz_tolerance[0] &&
                   std::fabs(p[1] - goal.tgt_off_pose[1]) <= goal.xyz_tolerance[1] &&
                   std::fabs(p[2] - goal.tgt_off_pose[2]) <= goal.xyz_tolerance[2];
        }
        return false;
    }
    // :1511-1580 (single-waypoint actions: limits, then parent -> wp0)
    int checkAction(const std::vector<double>& state, const std::vector<double>& wp) const
    {
        if (!robot->checkJointLimits(wp)) return 0x20;
        if (!checker->isStateToStateValid(state, wp)) return 0x40;
        return 0;
    }
    // :1174-1206 + bfs_heuristic.cpp:148-163
    int GetGoalHeuristic(int state_id) const
    {
        double p[3];
        if (state_id == goal_state_id) {
            p[0] = goal.tgt_off_pose[0]; p[1] = goal.tgt_off_pose[1]; p[2] = goal.tgt_off_pose[2];
        } else {
            robot->computePlanningLinkFK(state_angles[state_id], p);
        }
        return heur->heuristicAtPoint(p);
    }
    bool setGoal(const GoalConstraint& g)
    {
        goal = g;
        heur->updateGoal(goal);  // observer notification, robot_planning_space.cpp:96-131
        return true;
    }
    // :1944-1981
    bool setStart(const std::vector<double>& state)
    {
        if ((int)state.size() < robot->jointVariableCount()) return false;
        if (!robot->checkJointLimits(state)) return false;
        if (!checker->isStateValid(state)) return false;
        std::vector<int> c;
        stateToCoord(state, c);
        start_state_id = getOrCreateState(c, state);
        return true;
    }
    // :219-313 with manip_lattice_action_space.cpp:376-449 inlined
    void GetSuccs(int state_id, std::vector<int>* succs, std::vector<int>* costs)
    {
        if (trace) trace->clear();
        if (state_id == goal_state_id) return;
        ++expansions;
        const std::vector<double> parent = state_angles[state_id];  // copy: the table may grow
        double pose[3];
        robot->computePlanningLinkFK(parent, pose);
        const double goal_dist = heur->getMetricGoalDistance(pose[0], pose[1], pose[2]);
        std::vector<double> wp;
        std::vector<int> succ_coord;
        for (size_t pi = 0; pi < actions->mprims.size(); ++pi) {
            const MotionPrimitive& prim = actions->mprims[pi];
            SuccessorRecord rec;
            rec.prim = (int)pi;
            if (!actions->mprimActive(goal_dist, prim.type)) {
                rec.flags = 0x10;
                if (trace) trace->push_back(rec);
                continue;
            }
            if (prim.type == LONG_DISTANCE || prim.type == SHORT_DISTANCE) {
                if (!actions->applyMotionPrimitive(parent, prim, wp)) {
                    rec.flags = 0x10;
                    if (trace) trace->push_back(rec);
                    continue;
                }
            } else if (prim.type == SNAP_TO_XYZ_RPY && goal.type == JOINT_STATE_GOAL) {
                wp = goal.angles;  // :551-559
            } else {
                rec.flags = 0x10;  // IK snaps: out of scope (no IK)
                if (trace) trace->push_back(rec);
                continue;
            }
            ++succ_evals;
            const long l0 = checker->grid->lookups;
            const int viol = checkAction(parent, wp);
            rec.lookups = checker->grid->lookups - l0;
            rec.q = wp;
            if (viol) {
                rec.flags = viol;
                if (trace) trace->push_back(rec);
                continue;
            }
            stateToCoord(wp, succ_coord);
            const int succ_id = getOrCreateState(succ_coord, wp);
            const bool is_goal = isGoal(wp);
            succs->push_back(is_goal ? goal_state_id : succ_id);
            const int c = (int)(1000 * prim.weight);  // :1414-1437 (DefaultCostMultiplier*actionWeight)
            costs->push_back(c);
            rec.flags = 1 | (is_goal ? 2 : 0);
            rec.coord = succ_coord;
            rec.cost = c;
            if (trace) {
                double p[3];
                robot->computePlanningLinkFK(wp, p);
                rec.h = heur->heuristicAtPoint(p);
                trace->push_back(rec);
            }
        }
    }
};

// ---------------------------------------------------------------------------
// path post-processing: smpl/src/post_processing.cpp:52-67 (distance), 100-127 (joint-space
// shortcut generator), 281-309 (ShortcutPath, JOINT_SPACE), 430-445 (costs), 464-523 (InterpolatePath);
// the generic loop of smpl/include/smpl/geometry/detail/shortcut.hpp:110-286 with one generator and
// granularity 1 (shortcut.h:98-99 defaults, comparator std::less_equal); PlannerInterface::postProcessPath
// smpl_ros/src/ros/planner_interface.cpp:2651-2697.
// ---------------------------------------------------------------------------

// The loop of sbpl::shortcut::ShortcutPath (smpl/include/smpl/geometry/detail/shortcut.hpp:110-286) over point INDICES,
// one generator that returns the two-point path {start, end} (post_processing.cpp:100-127), window 1, granularity 1,
// comparator std::less_equal.  seg[i] = cost of the original transition i -> i+1.  Pinned against the reference's own
// header compiled in place (oracle/_ref/shortcut_ref, tests/golden/shortcut_ref.json).
template <class Generate>
inline void shortcut_indices(size_t psize, const std::vector<double>& seg, Generate generate, std::vector<int>& out)
{
    out.clear();
    if (psize == 0) return;
    if (psize < 2) { out.push_back(0); return; }
    std::vector<double> accum(psize);
    accum[0] = 0.0;
    for (size_t i = 1; i < psize; ++i) accum[i] = accum[i - 1] + seg[i - 1];
    const size_t granularity = 1;
    size_t start = 0, end = std::min(psize - 1, granularity);
    // the best path of the segment of interest: either the original points [start, end] or a direct connection
    bool best_direct = false;
    size_t best_last = end;            // last original index the best path reaches
    double best_cost = accum[end] - accum[start];
    double cost = 0.0;
    if (generate(start, end, cost) && cost <= best_cost) { best_direct = true; best_cost = cost; }
    out.push_back(0);
    auto emit_best = [&]() {
        if (best_direct) out.push_back((int)best_last);
        else for (size_t i = start + 1; i <= best_last; ++i) out.push_back((int)i);
    };
    while (end != psize) {
        bool improved = false;
        const size_t look_dist = std::min(granularity, psize - end - 1);   // distance(curr_end, plast) - 1
        if (look_dist != 0) {
            const double exp_cost = accum[end + look_dist] - accum[end];
            double new_cost = best_cost + exp_cost;
            if (generate(start, end + look_dist, cost) && cost <= new_cost) {
                improved = true;
                best_direct = true;
                best_last = end + look_dist;
                new_cost = cost;
            }
            best_cost = new_cost;
        }
        if (improved) {
            end += look_dist;
        } else if (look_dist == 0) {
            end = psize;
        } else {
            emit_best();
            start = end;
            end += look_dist;
            best_direct = false;
            best_last = end;
            best_cost = accum[end] - accum[start];
            if (generate(start, end, cost) && cost <= best_cost) { best_direct = true; best_cost = cost; }
        }
    }
    emit_best();
}

struct PostProcessor {
    PlanningRobotModel* robot;
    CollisionSpace* cc;
    bool fork_interpolate_limits_bug = true;   // [FORK] collision_space.cpp:592-597 rejects paths that ARE within limits
    long edge_checks = 0, state_checks = 0;

    // post_processing.cpp:52-67
    double distance(const std::vector<double>& from, const std::vector<double>& to) const
    {
        double dist = 0.0;
        for (size_t v = 0; v < from.size(); ++v) {
            if (!robot->bounded[v]) dist += shortest_angle_dist(to[v], from[v]);
            else dist += std::fabs(to[v] - from[v]);
        }
        return dist;
    }
    // JointPositionShortcutPathGenerator (:100-127)
    bool generate(const std::vector<double>& a, const std::vector<double>& b, double& cost)
    {
        ++edge_checks;
        if (!cc->isStateToStateValid(a, b)) return false;
        cost = distance(a, b);
        return true;
    }
    // shortcut.hpp:110-286
    void shortcutPath(const std::vector<std::vector<double>>& pin, std::vector<std::vector<double>>& pout)
    {
        pout.clear();
        const size_t psize = pin.size();
        if (psize < 2) { pout = pin; return; }
        std::vector<double> seg(psize - 1);
        for (size_t i = 0; i + 1 < psize; ++i) seg[i] = distance(pin[i], pin[i + 1]);   // ComputePositionPathCosts (:430-445)
        std::vector<int> idx;
        shortcut_indices(psize, seg, [&](size_t a, size_t b, double& cost) { return generate(pin[a], pin[b], cost); }, idx);
        for (int i : idx) pout.push_back(pin[i]);
    }
    // collision_space.cpp:583-612 + 776-793
    bool withinLimits(const std::vector<double>& q) const
    {
        for (size_t v = 0; v < q.size(); ++v) {
            if (!(robot->continuous[v] || !robot->bounded[v] || (q[v] >= robot->min_limits[v] && q[v] <= robot->max_limits[v]))) return false;
        }
        return true;
    }
    bool ccInterpolate(const std::vector<double>& a, const std::vector<double>& b, std::vector<std::vector<double>>& out) const
    {
        const bool wa = withinLimits(a), wb = withinLimits(b);
        if (fork_interpolate_limits_bug ? (wa || wb) : (!wa || !wb)) return false;
        const int W = cc->waypointCount(a, b);
        std::vector<double> diffs;
        cc->interpolationDiffs(a, b, diffs);
        const double inv = W > 0 ? 1.0 / (double)(W - 1) : 0.0;
        out.assign(W, std::vector<double>(a.size()));
        for (int n = 0; n < W; ++n) {
            const double alpha = (double)n * inv;
            for (size_t v = 0; v < a.size(); ++v) out[n][v] = a[v] + alpha * diffs[v];
        }
        return true;
    }
    // post_processing.cpp:464-523
    bool interpolatePath(std::vector<std::vector<double>>& path)
    {
        if (path.empty()) return true;
        std::vector<std::vector<double>> opath;
        opath.push_back(path.front());
        for (size_t i = 0; i + 1 < path.size(); ++i) {
            std::vector<std::vector<double>> ipath;
            if (!ccInterpolate(path[i], path[i + 1], ipath)) return false;
            bool collision = false;
            for (const auto& pt : ipath) {
                ++state_checks;
                if (!cc->isStateValid(pt)) { collision = true; break; }
            }
            if (collision) { opath.push_back(path[i + 1]); continue; }
            if (!ipath.empty()) opath.insert(opath.end(), ipath.begin() + 1, ipath.end());
        }
        path = std::move(opath);
        return true;
    }
    // planner_interface.cpp:2651-2697 (shortcut_path and interpolate_path both requested)
    void postProcessPath(std::vector<std::vector<double>>& path, bool shortcut, bool interpolate)
    {
        if (shortcut) {
            (void)interpolatePath(path);   // on failure the path is left as it was (:2661-2666)
            std::vector<std::vector<double>> ipath = path;
            shortcutPath(ipath, path);
        }
        if (interpolate) (void)interpolatePath(path);
    }
};

// ---------------------------------------------------------------------------
// intrusive_heap (smpl/include/smpl/detail/intrusive_heap.hpp:145-166, 346-395)
// Index 0 is unused; element positions are kept in the elements themselves.
// PINNED by oracle/_ref/heap_ref (the reference header compiled in place).
// ---------------------------------------------------------------------------

template <class T, class Less>
struct IntrusiveHeap {
    std::vector<T*> data;
    Less less;
    IntrusiveHeap() : data(1, nullptr) {}
    bool empty() const { return data.size() == 1; }
    size_t size() const { return data.size() - 1; }
    T* min() const { return data[1]; }
    bool contains(T* e) const { return e->heap_index != 0; }
    void clear()
    {
        for (size_t i = 1; i < data.size(); ++i) data[i]->heap_index = 0;
        data.resize(1);
    }
    void push(T* e)
    {
        e->heap_index = data.size();
        data.push_back(e);
        percolate_up(data.size() - 1);
    }
    void pop()
    {
        data[1]->heap_index = 0;
        data[1] = data.back();
        data.pop_back();
        percolate_down(1);
    }
    void decrease(T* e) { percolate_up(e->heap_index); }
    void increase(T* e) { percolate_down(e->heap_index); }
    void erase(T* e)
    {
        const size_t pos = e->heap_index;
        data[pos] = data.back();
        data[pos]->heap_index = pos;
        e->heap_index = 0;
        data.pop_back();
        percolate_down(pos);
    }
    void make()
    {
        for (size_t i = (data.size() - 1) >> 1; i >= 1; --i) percolate_down(i);
    }
    void percolate_down(size_t pivot)
    {
        if (pivot >= data.size()) return;
        size_t left = pivot << 1, right = (pivot << 1) + 1;
        T* tmp = data[pivot];
        while (left < data.size()) {
            size_t s = right;
            if (right >= data.size() || less(*data[left], *data[right])) s = left;
            if (less(*data[s], *tmp)) {
                data[pivot] = data[s];
                data[pivot]->heap_index = pivot;
                pivot = s;
            } else {
                break;
            }
            left = pivot << 1;
            right = (pivot << 1) + 1;
        }
        data[pivot] = tmp;
        data[pivot]->heap_index = pivot;
    }
    void percolate_up(size_t pivot)
    {
        T* tmp = data[pivot];
        while (pivot != 1) {
            const size_t p = pivot >> 1;
            if (less(*data[p], *tmp)) break;
            data[pivot] = data[p];
            data[pivot]->heap_index = pivot;
            pivot = p;
        }
        data[pivot] = tmp;
        data[pivot]->heap_index = pivot;
    }
};

// ---------------------------------------------------------------------------
// ARA* (smpl/src/search/arastar.cpp:107-215 replan, 486-527 improvePath,
// 531-568 expand, 570-582 reorderOpen/computeKey, 584-627 state init)
// ---------------------------------------------------------------------------

static const unsigned int INFINITECOST = 1000000000;  // SBPL sbpl/config.h (third party, not in tree: unpinned)

struct ARAStar {
    struct SearchState {
        size_t heap_index = 0;
        int state_id = 0;
        unsigned int g = 0, h = 0, f = 0, eg = 0;
        unsigned short iteration_closed = 0, call_number = 0;
        SearchState* bp = nullptr;
        bool incons = false;
    };
    struct Less { bool operator()(const SearchState& a, const SearchState& b) const { return a.f < b.f; } };

    ManipLattice* space;
    double initial_eps = 1.0, final_eps = 1.0, delta_eps = 1.0;
    bool improve = true;
    bool bounded = false;           // expansion bound (the TIME bound is nondeterministic and not restated)
    int max_expansions_init = 0, max_expansions = 0;
    std::vector<SearchState*> states;
    int start_state_id = -1, goal_state_id = -1;
    IntrusiveHeap<SearchState, Less> open;
    std::vector<SearchState*> incons;
    double curr_eps = 1.0;
    int iteration = 1;
    int call_number = 0;
    int last_start_state_id = -1, last_goal_state_id = -1;
    int expand_count_init = 0, expand_count = 0;
    double satisfied_eps = std::numeric_limits<double>::infinity();
    std::vector<int> expansion_log;  // state ids in expansion order (parity evidence)
    bool log_expansions = true;

    explicit ARAStar(ManipLattice* s) : space(s) {}
    ~ARAStar() { for (SearchState* s : states) delete s; }

    enum { SUCCESS = 0, PARTIAL_SUCCESS, START_NOT_SET, GOAL_NOT_SET, TIMED_OUT, EXHAUSTED_OPEN_LIST };

    int set_start(int id) { start_state_id = id; return 1; }
    int set_goal(int id) { goal_state_id = id; return 1; }
    void force_planning_from_scratch() { last_start_state_id = -1; last_goal_state_id = -1; }

    SearchState* getSearchState(int id)
    {
        if ((int)states.size() <= id) states.resize(id + 1, nullptr);
        if (!states[id]) {
            states[id] = new SearchState;
            states[id]->state_id = id;
            states[id]->call_number = 0;
        }
        return states[id];
    }
    void reinitSearchState(SearchState* s)
    {
        if (s->call_number != call_number) {
            s->g = INFINITECOST;
            s->h = (unsigned int)space->GetGoalHeuristic(s->state_id);
            s->f = INFINITECOST;
            s->eg = INFINITECOST;
            s->iteration_closed = 0;
            s->call_number = (unsigned short)call_number;
            s->bp = nullptr;
            s->incons = false;
        }
    }
    // arastar.cpp:579-582.  (unsigned)(double) beyond UINT_MAX is undefined in
    // C++; x86-64 GCC converts through a 64-bit integer, restated explicitly.
    unsigned int computeKey(const SearchState* s) const
    {
        return s->g + (unsigned int)(long long)(curr_eps * s->h);
    }
    void reorderOpen()
    {
        for (size_t i = 1; i < open.data.size(); ++i) open.data[i]->f = computeKey(open.data[i]);
        open.make();
    }
    bool timedOut(int elapsed_expansions) const
    {
        if (!bounded) return false;
        if (satisfied_eps == std::numeric_limits<double>::infinity()) return elapsed_expansions >= max_expansions_init;
        return elapsed_expansions >= max_expansions;
    }
    void expand(SearchState* s)
    {
        std::vector<int> succs, costs;
        space->GetSuccs(s->state_id, &succs, &costs);
        for (size_t i = 0; i < succs.size(); ++i) {
            SearchState* ss = getSearchState(succs[i]);
            reinitSearchState(ss);
            const int new_cost = (int)(s->eg + costs[i]);
            if ((unsigned int)new_cost < ss->g) {  // int vs unsigned compare, :546-548
                ss->g = new_cost;
                ss->bp = s;
                if (ss->iteration_closed != iteration) {
                    ss->f = computeKey(ss);
                    if (open.contains(ss)) open.decrease(ss);
                    else open.push(ss);
                } else if (!ss->incons) {
                    incons.push_back(ss);  // incons is never set true (:563-565)
                }
            }
        }
    }
    int improvePath(SearchState* goal_state, int& elapsed_expansions)
    {
        while (!open.empty()) {
            SearchState* min_state = open.min();
            if (min_state->f >= goal_state->f || min_state == goal_state) return SUCCESS;
            if (timedOut(elapsed_expansions)) return TIMED_OUT;
            open.pop();
            min_state->iteration_closed = (unsigned short)iteration;
            min_state->eg = min_state->g;
            if (log_expansions) expansion_log.push_back(min_state->state_id);
            expand(min_state);
            ++elapsed_expansions;
        }
        return EXHAUSTED_OPEN_LIST;
    }
    // returns 1 on success like the reference (!code)
    int replan(std::vector<int>* solution, int* cost)
    {
        if (start_state_id < 0) return !START_NOT_SET;
        if (goal_state_id < 0) return !GOAL_NOT_SET;
        SearchState* start_state = getSearchState(start_state_id);
        SearchState* goal_state = getSearchState(goal_state_id);
        if (start_state_id != last_start_state_id) {
            open.clear();
            incons.clear();
            ++call_number;
            reinitSearchState(start_state);
            reinitSearchState(goal_state);
            start_state->g = 0;
            start_state->f = computeKey(start_state);
            open.push(start_state);
            iteration = 1;
            expand_count_init = 0;
            expand_count = 0;
            curr_eps = initial_eps;
            satisfied_eps = std::numeric_limits<double>::infinity();
            last_start_state_id = start_state_id;
        }
        if (goal_state_id != last_goal_state_id) {
            for (SearchState* s : states) if (s) s->h = (unsigned int)space->GetGoalHeuristic(s->state_id);
            reorderOpen();
            last_goal_state_id = goal_state_id;
        }
        int num_expansions = 0;
        int err = SUCCESS;
        while (satisfied_eps > final_eps) {
            if (curr_eps == satisfied_eps) {
                if (!improve) break;
                ++iteration;
                curr_eps -= delta_eps;
                curr_eps = std::max(curr_eps, final_eps);
                for (SearchState* s : incons) {
                    s->incons = false;
                    open.push(s);
                }
                reorderOpen();
                incons.clear();
            }
            err = improvePath(goal_state, num_expansions);
            if (curr_eps == initial_eps) expand_count_init += num_expansions;
            if (err) break;
            satisfied_eps = curr_eps;
        }
        expand_count += num_expansions;
        if (satisfied_eps == std::numeric_limits<double>::infinity()) return !err;
        for (SearchState* s = goal_state; s; s = s->bp) solution->push_back(s->state_id);
        std::reverse(solution->begin(), solution->end());
        *cost = (int)goal_state->g;
        return !SUCCESS;
    }
};

}  // namespace oracle
