// smpl_amd/csrc/kernels.hip -- gfx950 (CDNA4, wave64) kernels of the ARA* state-expansion path.
//
//   k_state_prep      one thread per open state: planning-link FK -> metric goal distance
//                     (manip_lattice_action_space.cpp:385-397) and the validity of the state
//                     itself, which is waypoint 0 of every outgoing edge (collision_space.cpp:561-577)
//   k_expand          one thread per (open state, motion primitive): gating, primitive application,
//                     joint limits, edge collision check, discretisation, goal test, heuristic, cost
//                     (manip_lattice.cpp:254-305 loop body)
//   k_edge_valid      CollisionChecker::isStateToStateValid for a batch of edges
//   k_state_valid     CollisionChecker::isStateValid for a batch of states
//   k_heuristic       BfsHeuristic::GetGoalHeuristic for a batch of states
//   k_bfs_*           level-synchronous 26-connected BFS (bfs3d.cpp:507-547)
//
// No MFMA: the path is integer/byte gathers from the voxel grid plus a short serial FK chain in fp64.
// The sphere trees are staged in LDS; per-thread scratch (tree-root positions, saved link transforms,
// DFS stack) lives in LDS in structure-of-arrays form (conflict-free: lane i touches word i).
// Compile with -ffp-contract=off (arithmetic contract, det_math.h).
#ifndef __HIPCC_RTC__   // hiprtc (per-robot specialisation, specialize.cpp) brings its own runtime declarations
#include <hip/hip_runtime.h>
#endif

#include "det_math.h"
#include "device_types.h"
#include "kernels.h"

#define BLOCK SMPLX_BLOCK

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------

// LDS pointers are declared in address space 3: 32-bit, always lowered to ds_* instructions, half the
// scalar-register cost of generic pointers (the collision kernels are SGPR-bound).
#define LDS_AS __attribute__((address_space(3)))
typedef const LDS_AS SmplxJoint* JointPtr;
typedef const LDS_AS SmplxNode* NodePtr;
typedef const LDS_AS int* IntPtr;
typedef const LDS_AS double* DblPtr;

// The compiled model as the kernels see it: counts plus pointers into the LDS copy of the packed model
// (device_types.h SMPLX_BH_*).  Field names match SmplxModelDev so the device code reads the same either way.
struct ModelLds {
    int njoints, nvars, ntrees, nnodes, npairs, nslots, nroot;
    JointPtr joints;
    NodePtr nodes;
    IntPtr tree_first, tree_joint, tree_root_slot, pair_first, pair_other;
    DblPtr var_min, var_max, var_min_norm, var_k, coord_delta;
    IntPtr coord_vals, var_type;
};

struct ThreadLds {
    NodePtr nodes;            // shared: sphere trees
    LDS_AS double* d;         // per-thread doubles, SoA: d[e * BLOCK + tid]
    LDS_AS unsigned char* stk;   // per-thread byte stack, SoA
    int root_base;            // first double of root positions (3 per tree)
    int slot_base;            // first double of saved transforms (12 per slot)
    int q_base;               // first double of the configuration's joint values (one per planning variable)
    int stride;               // threads per block (SoA stride)
};

__device__ __forceinline__ LDS_AS double& lds_d(const ThreadLds& L, int e) { return L.d[e * L.stride + threadIdx.x]; }
__device__ __forceinline__ LDS_AS unsigned char& lds_b(const ThreadLds& L, int e) { return L.stk[e * L.stride + threadIdx.x]; }

// Per-variable model data.  In the per-robot build (SMPLX_CONST_MODEL) these are literals and the loops over the
// variables unroll; the generic kernels read the LDS copy of the model.
#ifdef SMPLX_CONST_MODEL
#include SMPLX_CONST_MODEL
#define MV_NVARS(M) CM_NV
#define MV_TYPE(M, v) CM_VAR_TYPE[v]
#define MV_MIN(M, v) CM_VAR_MIN[v]
#define MV_MAX(M, v) CM_VAR_MAX[v]
#define MV_MIN_NORM(M, v) CM_VAR_MIN_NORM[v]
#define MV_K(M, v) CM_VAR_K[v]
#define MV_COORD_DELTA(M, v) CM_COORD_DELTA[v]
#define MV_COORD_VALS(M, v) CM_COORD_VALS[v]
#define MV_UNROLL _Pragma("unroll")
#else
#define MV_NVARS(M) (M)->nvars
#define MV_TYPE(M, v) (M)->var_type[v]
#define MV_MIN(M, v) (M)->var_min[v]
#define MV_MAX(M, v) (M)->var_max[v]
#define MV_MIN_NORM(M, v) (M)->var_min_norm[v]
#define MV_K(M, v) (M)->var_k[v]
#define MV_COORD_DELTA(M, v) (M)->coord_delta[v]
#define MV_COORD_VALS(M, v) (M)->coord_vals[v]
#define MV_UNROLL
#endif

// p = T * c   (robot_collision_state.h:576); ((a*x + b*y) + c*z) + t
__device__ __forceinline__ void xform(const double T[12], const double c[3], double p[3])
{
    p[0] = ((T[0] * c[0] + T[1] * c[1]) + T[2] * c[2]) + T[3];
    p[1] = ((T[4] * c[0] + T[5] * c[1]) + T[6] * c[2]) + T[7];
    p[2] = ((T[8] * c[0] + T[9] * c[1]) + T[10] * c[2]) + T[11];
}

// local transform of a joint: origin * R(q)   (transform_functions.h:95-258)
__device__ __forceinline__ void joint_matrix(JointPtr j, double q, double J[12])
{
    DblPtr o = j->origin;
    const int kind = j->kind;
    if (kind == SMPLX_TK_FIXED) {
#pragma unroll
        for (int i = 0; i < 12; ++i) J[i] = o[i];
        return;
    }
    if (kind == SMPLX_TK_PRISMATIC) {   // translates along local Z whatever the axis (:218-226)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            J[4 * i + 0] = o[4 * i + 0]; J[4 * i + 1] = o[4 * i + 1]; J[4 * i + 2] = o[4 * i + 2];
            J[4 * i + 3] = ((o[4 * i + 0] * 0.0 + o[4 * i + 1] * 0.0) + o[4 * i + 2] * q) + o[4 * i + 3];
        }
        return;
    }
    double s, c;
    smplx_sincos(q, &s, &c);
    if (kind == SMPLX_TK_REV_X) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            J[4 * i + 0] = o[4 * i + 0];
            J[4 * i + 1] = c * o[4 * i + 1] + s * o[4 * i + 2];
            J[4 * i + 2] = c * o[4 * i + 2] - s * o[4 * i + 1];
            J[4 * i + 3] = o[4 * i + 3];
        }
    } else if (kind == SMPLX_TK_REV_Y) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            J[4 * i + 0] = c * o[4 * i + 0] - s * o[4 * i + 2];
            J[4 * i + 1] = o[4 * i + 1];
            J[4 * i + 2] = s * o[4 * i + 0] + c * o[4 * i + 2];
            J[4 * i + 3] = o[4 * i + 3];
        }
    } else if (kind == SMPLX_TK_REV_Z) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            J[4 * i + 0] = o[4 * i + 0] * c + o[4 * i + 1] * s;
            J[4 * i + 1] = o[4 * i + 1] * c - o[4 * i + 0] * s;
            J[4 * i + 2] = o[4 * i + 2];
            J[4 * i + 3] = o[4 * i + 3];
        }
    } else {   // generic axis: o * AngleAxis(q, axis)  (Eigen toRotationMatrix restated)
        const double ax = j->axis[0], ay = j->axis[1], az = j->axis[2];
        const double sx = s * ax, sy = s * ay, sz = s * az;
        const double c1 = 1.0 - c;
        const double cx = c1 * ax, cy = c1 * ay, cz = c1 * az;
        double R[9];
        double tmp;
        tmp = cx * ay; R[1] = tmp - sz; R[3] = tmp + sz;
        tmp = cx * az; R[2] = tmp + sy; R[6] = tmp - sy;
        tmp = cy * az; R[5] = tmp - sx; R[7] = tmp + sx;
        R[0] = cx * ax + c; R[4] = cy * ay + c; R[8] = cz * az + c;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
                J[4 * i + k] = (o[4 * i + 0] * R[k] + o[4 * i + 1] * R[3 + k]) + o[4 * i + 2] * R[6 + k];
            J[4 * i + 3] = o[4 * i + 3];
        }
    }
}

// T = T * J   (robot_collision_state.h:419-421)
__device__ __forceinline__ void mul_affine(double T[12], const double J[12])
{
    double R[12];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            R[4 * i + k] = (T[4 * i + 0] * J[k] + T[4 * i + 1] * J[4 + k]) + T[4 * i + 2] * J[8 + k];
        R[4 * i + 3] = ((T[4 * i + 0] * J[3] + T[4 * i + 1] * J[7]) + T[4 * i + 2] * J[11]) + T[4 * i + 3];
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = R[i];
}

// the identity-origin forms (SMPLX_TK_*_T) with the origin's translation already in registers
__device__ __forceinline__ void apply_joint_t(int kind, double tx, double ty, double tz, double q, double T[12], bool on_root)
{
    if (on_root) {
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = 0.0;
        T[0] = 1.0; T[5] = 1.0; T[10] = 1.0;
        T[3] = tx; T[7] = ty; T[11] = tz;
        if (kind == SMPLX_TK_FIXED_T) return;
        double s, c;
        smplx_sincos(q, &s, &c);
        if (kind == SMPLX_TK_REV_X_T) { T[5] = c; T[6] = 0.0 - s; T[9] = s; T[10] = c; }
        else if (kind == SMPLX_TK_REV_Y_T) { T[0] = c; T[2] = s; T[8] = 0.0 - s; T[10] = c; }
        else { T[0] = c; T[1] = 0.0 - s; T[4] = s; T[5] = c; }
        return;
    }
    // translation first: it uses the rotation of T before it is rotated
    const double n3 = ((T[0] * tx + T[1] * ty) + T[2] * tz) + T[3];
    const double n7 = ((T[4] * tx + T[5] * ty) + T[6] * tz) + T[7];
    const double n11 = ((T[8] * tx + T[9] * ty) + T[10] * tz) + T[11];
    T[3] = n3; T[7] = n7; T[11] = n11;
    if (kind == SMPLX_TK_FIXED_T) return;
    double s, c;
    smplx_sincos(q, &s, &c);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double a = T[4 * i + 0], b = T[4 * i + 1], d = T[4 * i + 2];
        if (kind == SMPLX_TK_REV_X_T) { T[4 * i + 1] = b * c + d * s; T[4 * i + 2] = d * c - b * s; }
        else if (kind == SMPLX_TK_REV_Y_T) { T[4 * i + 0] = a * c - d * s; T[4 * i + 2] = a * s + d * c; }
        else { T[4 * i + 0] = a * c + b * s; T[4 * i + 1] = b * c - a * s; }
    }
}

// One step of the kinematic chain: T = T * J(q), or T = J(q) for a joint on the root link.
// For origins whose rotation is exactly the identity (SMPLX_TK_*_T) the general form
//   J = origin * R_axis(q)   (transform_functions.h:104-207),   T' = T * J   (robot_collision_state.h:419-421)
// multiplies by 0 and 1 only; the terms x*1 and y*0 are exact, adding +-0 changes no non-zero value, and
// a*(-s) + b*c == b*c - a*s bit for bit, so the shortened expressions below give identical bits.
__device__ __forceinline__ void apply_joint(JointPtr jt, double q, double T[12], bool on_root)
{
    const int kind = jt->kind;
    if (kind < SMPLX_TK_FIXED_T) {
        double J[12];
        joint_matrix(jt, q, J);
        if (on_root) {
#pragma unroll
            for (int i = 0; i < 12; ++i) T[i] = J[i];
        } else {
            mul_affine(T, J);
        }
        return;
    }
    DblPtr o = jt->origin;
    apply_joint_t(kind, o[3], o[7], o[11], q, T, on_root);
}

#define SMPLX_GLOBAL_AS __attribute__((address_space(1)))
// A pointer read out of a struct in memory (or out of LDS) is a FLAT address to the compiler: its loads and stores count on
// the LDS counter as well as on the memory counter, so every wait for an LDS read behind them waits for HBM too, and
// the other way round.  The buffers of this engine are all device memory: as_global says so.  (The type has to carry it: a
// cast to address space 1 and back is folded away, and the compiler takes no hint from an assumption.)
template <class T>
__device__ __forceinline__ SMPLX_GLOBAL_AS T* as_global(T* p) { return (SMPLX_GLOBAL_AS T*)p; }

// voxel lookup: squared cell distance at a world point, 0 outside the grid
// (occupancy_grid.h:234 -> distance_map.hpp:281-300, 520-536)
__device__ __forceinline__ int grid_d2(const SmplxGridDev& g, const double p[3])
{
    const int x = (int)(g.inv_res * (p[0] - g.origin_minus_res[0]) + 0.5) - 1;
    const int y = (int)(g.inv_res * (p[1] - g.origin_minus_res[1]) + 0.5) - 1;
    const int z = (int)(g.inv_res * (p[2] - g.origin_minus_res[2]) + 0.5) - 1;
    // no branch around the load: a load inside a branch is waited for where the branches rejoin, which put the whole
    // round trip in front of whatever the caller meant to overlap with it.  Out of the grid: cell 0 is read and dropped.
    const bool outside = x < 0 || y < 0 || z < 0 || x >= g.n[0] || y >= g.n[1] || z >= g.n[2];
    const size_t brick = ((size_t)(x >> 2) * g.bricks[1] + (y >> 2)) * g.bricks[2] + (z >> 2);
    const size_t cell = brick * 64 + ((x & 3) << 4) + ((y & 3) << 2) + (z & 3);
    // the pointer comes out of a struct read from memory, so the compiler takes it for a FLAT address: a flat load counts
    // on the LDS counter as well, and every wait for an LDS read behind it (saved transforms, tree nodes) waited for the
    // grid gather too.  It is device memory: say so.
    const SMPLX_GLOBAL_AS unsigned short* d2 = (const SMPLX_GLOBAL_AS unsigned short*)g.d2;
    const int v = (int)d2[outside ? (size_t)0 : cell];
    return outside ? 0 : v;
}

// interpolated value of planning variable v on the edge start -> finish at parameter alpha
// (robot_motion_collision_model.h:221-247 diffs, 297-320 interpolate)
__device__ __forceinline__ double edge_diff(const ModelLds* __restrict__ M, int v, double sv, double fv)
{
    return (MV_TYPE(M, v) == SMPLX_JT_CONTINUOUS) ? smplx_shortest_angle_diff(fv, sv) : fv - sv;
}

// sphere tree vs voxel grid for the tree on the current link (collision_operations.h:105-164).
// Returns false at the first colliding leaf.  The root position comes back for the sphere-sphere tests.
__device__ __forceinline__ bool check_tree(const ModelLds* __restrict__ M, const ThreadLds& L, const SmplxGridDev& g,
                                           int t, const double T[12], int& lookups, double root_p[3])
{
    const int root = M->tree_first[t + 1] - 1;
    int sp = 0;
    int node = root;
    while (true) {
        const LDS_AS SmplxNode& nd = L.nodes[node];
        double c[3] = {nd.c[0], nd.c[1], nd.c[2]};
        double p[3];
        xform(T, c, p);
        if (node == root) { root_p[0] = p[0]; root_p[1] = p[1]; root_p[2] = p[2]; }
        ++lookups;
#ifdef ABL_NO_LOOKUP
        const int d2 = 60000 + (int)(p[0] * 0.0);
#else
        const int d2 = grid_d2(g, p);
#endif
        if (d2 < nd.thr) {              // CheckSphereCollision fails (collision_operations.h:67-77)
            if (nd.left < 0) return false;
            const double rl = L.nodes[nd.left].r, rr = L.nodes[nd.right].r;
            // larger child is examined first (:150-156): push the other one
            if (rl > rr) { lds_b(L, sp++) = (unsigned char)nd.right; node = nd.left; }
            else { lds_b(L, sp++) = (unsigned char)nd.left; node = nd.right; }
            continue;
        }
        if (sp == 0) break;
        node = lds_b(L, --sp);
    }
    return true;
}

// value source for the configuration being checked
struct EdgeRef {
    const double* __restrict__ start;    // N doubles
    const double* __restrict__ finish;   // N doubles
    double alpha;
};

// joint values of the configuration on the edge at parameter alpha
// (robot_motion_collision_model.h:221-247 diffs, 297-320 interpolate), staged into per-thread LDS once per
// configuration (the slow path of the sphere-sphere pass re-reads them)
__device__ __forceinline__ void stage_config(const ModelLds* __restrict__ M, const ThreadLds& L, const EdgeRef& e)
{
    const int nv = MV_NVARS(M);
    MV_UNROLL
    for (int v = 0; v < nv; ++v) {
        const double sv = e.start[v];
        double q = sv;
        if (e.alpha != 0.0) q = sv + e.alpha * edge_diff(M, v, sv, e.finish[v]);   // start + 0*diff == start exactly
        lds_d(L, L.q_base + v) = q;
    }
}

__device__ __forceinline__ double config_var(const ModelLds* __restrict__ M, const ThreadLds& L, int v)
{
    (void)M;
    return lds_d(L, L.q_base + v);
}

// link transforms of two trees' links for one configuration (slow path of the sphere-sphere pass)
__device__ __forceinline__ void fk_two_links(const ModelLds* __restrict__ M, const ThreadLds& L, const EdgeRef& e,
                                          int ja, int jb, double Ta[12], double Tb[12])
{
    double T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = 0.0;
    const int last = ja > jb ? ja : jb;
    for (int j = 0; j <= last; ++j) {
        JointPtr jt = &M->joints[j];
        const double q = jt->var >= 0 ? config_var(M, L, jt->var) : 0.0;
        if (jt->src >= 0) {
#pragma unroll
            for (int i = 0; i < 12; ++i) T[i] = lds_d(L, L.slot_base + 12 * jt->src + i);
        }
        apply_joint(jt, q, T, jt->src == SMPLX_SRC_ROOT);
        if (jt->save_slot >= 0) {
#pragma unroll
            for (int i = 0; i < 12; ++i) lds_d(L, L.slot_base + 12 * jt->save_slot + i) = T[i];
        }
        if (j == ja) {
#pragma unroll
            for (int i = 0; i < 12; ++i) Ta[i] = T[i];
        }
        if (j == jb) {
#pragma unroll
            for (int i = 0; i < 12; ++i) Tb[i] = T[i];
        }
    }
}

// Checked link pairs whose root spheres overlap and are not both leaves wait here for the sphere-tree pass behind the
// chain: (earlier tree << 8 | later tree), 16 bits each, twelve of them in three words.  More than that -> every pair
// is rechecked.  (Four slots overflowed in most waves of random configurations, and a recheck of every pair pays a
// two-link FK per pair.)
#define SMPLX_PENDING_MAX 12
struct PendingPairs {
    unsigned long long w0, w1, w2;
    int n;
};
__device__ __forceinline__ bool pend_push(PendingPairs& P, int ta, int t)
{
    if (P.n >= SMPLX_PENDING_MAX) return false;
    const unsigned long long c = (unsigned long long)((ta << 8) | t) << (16 * (P.n & 3));
    const int k = P.n >> 2;
    if (k == 0) P.w0 |= c;
    else if (k == 1) P.w1 |= c;
    else P.w2 |= c;
    ++P.n;
    return true;
}
__device__ __forceinline__ int pend_get(const PendingPairs& P, int i)
{
    const int k = i >> 2;
    const unsigned long long w = k == 0 ? P.w0 : (k == 1 ? P.w1 : P.w2);
    return (int)((w >> (16 * (i & 3))) & 0xFFFF);
}

// the part of a joint record the chain step needs, in registers
struct JointHead {
    int kind, var, src, save_slot, tree;
    double tx, ty, tz, q;
};

__device__ __forceinline__ JointHead load_joint_head(const ModelLds* __restrict__ M, const ThreadLds& L, int j)
{
    JointPtr jt = &M->joints[j];
    JointHead h;
    h.kind = jt->kind; h.var = jt->var; h.src = jt->src; h.save_slot = jt->save_slot; h.tree = jt->tree;
    h.tx = jt->origin[3]; h.ty = jt->origin[7]; h.tz = jt->origin[11];
    h.q = h.var >= 0 ? lds_d(L, L.q_base + h.var) : 0.0;
    return h;
}

#ifdef SMPLX_CONST_MODEL
// ---------------------------------------------------------------------------------------------
// Per-robot specialisation: the chain structure (joint kinds, origins, which link carries which tree, the
// checked pairs) comes from compile-time constants (model_compile.cpp model_const_header), so the joint loop
// is straight-line code: no joint records read from LDS, no kind dispatch, root positions and joint values in
// registers.  Same operations in the same order as the generic path below: identical bits.
// ---------------------------------------------------------------------------------------------

struct ChainState {
    double T[12];
    double q[CM_NV];
    double roots[3 * (CM_NT > 0 ? CM_NT : 1)];   // only the slots that lead a pair are ever touched
    bool pair_hit, recheck_all;
    PendingPairs P;
    // the tree whose root lookup is in flight (software pipelining, see const_chain): its link transform and the
    // squared cell distance the lookup returns
    double Tp[12];
    int pd2;
    // RS (const_chain<.., true>): the saved link transforms here instead of in the thread's LDS slots -- 96 bytes of LDS per
    // thread and slot were what held the validity kernels at three waves per SIMD
    double S[SMPLX_MAX_SLOTS][12];
};

template <int T_, int K, int KEND>
__device__ __forceinline__ void const_pairs(const ThreadLds& L, ChainState& C, const double rp[3])
{
    if constexpr (K < KEND) {
        constexpr int ta = CM_PAIR_OTHER[K];
        constexpr int sa = CM_ROOT_SLOT[ta];
        constexpr bool a_first = ta < T_;
        const double ax = C.roots[3 * sa + 0], ay = C.roots[3 * sa + 1], az = C.roots[3 * sa + 2];
        const double dx = a_first ? rp[0] - ax : ax - rp[0];
        const double dy = a_first ? rp[1] - ay : ay - rp[1];
        const double dz = a_first ? rp[2] - az : az - rp[2];
        const double cd2 = (dx * dx + dy * dy) + dz * dz;
        constexpr double rr = a_first ? CM_ROOT_R[ta] + CM_ROOT_R[T_] : CM_ROOT_R[T_] + CM_ROOT_R[ta];
        if (!(cd2 > rr * rr)) {
            if constexpr (CM_ROOT_LEAF[ta] && CM_ROOT_LEAF[T_]) {
                C.pair_hit = true;
            } else {
                if (!pend_push(C.P, ta, T_)) C.recheck_all = true;
            }
        }
        const_pairs<T_, K + 1, KEND>(L, C, rp);
    }
}

// apply_joint_t with the origin's translation as literals: terms with an exactly-zero coefficient are dropped
// (x*0 is +-0 and adding it changes no non-zero value)
template <int J, bool OnRoot>
__device__ __forceinline__ void apply_joint_const(double q, double T[12])
{
    constexpr int kind = CM_KIND[J];
    constexpr double tx = CM_TX[J], ty = CM_TY[J], tz = CM_TZ[J];
    if constexpr (OnRoot) {
        apply_joint_t(kind, tx, ty, tz, q, T, true);
    } else {
        if constexpr (tx != 0.0 || ty != 0.0 || tz != 0.0) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double acc = 0.0;
                bool have = false;
                if constexpr (tx != 0.0) { acc = T[4 * i + 0] * tx; have = true; }
                if constexpr (ty != 0.0) { acc = have ? acc + T[4 * i + 1] * ty : T[4 * i + 1] * ty; have = true; }
                if constexpr (tz != 0.0) { acc = have ? acc + T[4 * i + 2] * tz : T[4 * i + 2] * tz; have = true; }
                T[4 * i + 3] = acc + T[4 * i + 3];
            }
        }
        if constexpr (kind != SMPLX_TK_FIXED_T) {
            double s, c;
            smplx_sincos(q, &s, &c);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double a = T[4 * i + 0], b = T[4 * i + 1], d = T[4 * i + 2];
                if constexpr (kind == SMPLX_TK_REV_X_T) { T[4 * i + 1] = b * c + d * s; T[4 * i + 2] = d * c - b * s; }
                else if constexpr (kind == SMPLX_TK_REV_Y_T) { T[4 * i + 0] = a * c - d * s; T[4 * i + 2] = a * s + d * c; }
                else { T[4 * i + 0] = a * c + b * s; T[4 * i + 1] = b * c - a * s; }
            }
        }
    }
}

// check_tree with the root sphere as literals, cut in two so that the root's grid lookup can be IN FLIGHT while the next
// joint of the chain is computed (every tree used to cost one exposed L2/HBM round trip: the compare-and-branch sat
// right behind its load).  issue_root computes the root position and starts the lookup; resolve_root, called after the
// next joint's arithmetic, looks at the answer: in free space the root clears and nothing is read from LDS; a root that
// does not clear hands over to the generic traversal at its children (larger child first, as check_tree does), with
// the link transform kept in C.Tp.  The order of the lookups -- and so the tally, also of a colliding configuration --
// is unchanged: tree k is resolved before tree k+1 is issued.
template <int T_>
__device__ __forceinline__ void issue_root(const SmplxGridDev& g, ChainState& C, int& lookups, double root_p[3])
{
    constexpr double cx = CM_ROOT_CX[T_], cy = CM_ROOT_CY[T_], cz = CM_ROOT_CZ[T_];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        // ((a*x + b*y) + c*z) + t with exactly-zero coefficients dropped (see apply_joint_const)
        double acc = 0.0;
        bool have = false;
        if constexpr (cx != 0.0) { acc = C.T[4 * i + 0] * cx; have = true; }
        if constexpr (cy != 0.0) { acc = have ? acc + C.T[4 * i + 1] * cy : C.T[4 * i + 1] * cy; have = true; }
        if constexpr (cz != 0.0) { acc = have ? acc + C.T[4 * i + 2] * cz : C.T[4 * i + 2] * cz; have = true; }
        root_p[i] = have ? acc + C.T[4 * i + 3] : C.T[4 * i + 3];
    }
    ++lookups;
#ifdef ABL_NO_LOOKUP
    C.pd2 = 60000 + (int)(root_p[0] * 0.0);
#else
    C.pd2 = grid_d2(g, root_p);
#endif
    if constexpr (CM_ROOT_LEFT[T_] >= 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) C.Tp[i] = C.T[i];   // only a tree that can be descended into needs its transform later
    }
}

template <int T_>
__device__ __forceinline__ bool resolve_root(const ModelLds* __restrict__ M, const ThreadLds& L, const SmplxGridDev& g,
                                             const double* Tp, int pd2, int& lookups)
{
    (void)M;
    if (!(pd2 < CM_ROOT_THR[T_])) return true;
    if constexpr (CM_ROOT_LEFT[T_] < 0) {
        return false;
    } else {
        // descend: the same loop as check_tree, entered below the root
        int sp = 0;
        int node;
        {
            const double rl = L.nodes[CM_ROOT_LEFT[T_]].r, rr = L.nodes[CM_ROOT_RIGHT[T_]].r;
            if (rl > rr) { lds_b(L, sp++) = (unsigned char)CM_ROOT_RIGHT[T_]; node = CM_ROOT_LEFT[T_]; }
            else { lds_b(L, sp++) = (unsigned char)CM_ROOT_LEFT[T_]; node = CM_ROOT_RIGHT[T_]; }
        }
        while (true) {
            const LDS_AS SmplxNode& nd = L.nodes[node];
            double c[3] = {nd.c[0], nd.c[1], nd.c[2]};
            double p[3];
            xform(Tp, c, p);
            ++lookups;
#ifdef ABL_NO_LOOKUP
            const int dd = 60000 + (int)(p[0] * 0.0);
#else
            const int dd = grid_d2(g, p);
#endif
            if (dd < nd.thr) {
                if (nd.left < 0) return false;
                const double rl = L.nodes[nd.left].r, rr = L.nodes[nd.right].r;
                if (rl > rr) { lds_b(L, sp++) = (unsigned char)nd.right; node = nd.left; }
                else { lds_b(L, sp++) = (unsigned char)nd.left; node = nd.right; }
                continue;
            }
            if (sp == 0) break;
            node = lds_b(L, --sp);
        }
        return true;
    }
}

// PT = the tree whose root lookup was issued at an earlier joint and has not been looked at yet (-1: none)
template <int J, int PT, bool RS = false>
__device__ __forceinline__ bool const_chain(const ModelLds* __restrict__ M, const ThreadLds& L, const SmplxGridDev& g,
                                            ChainState& C, int& lookups)
{
    if constexpr (J < CM_NJ) {
        constexpr int kind = CM_KIND[J], var = CM_VAR[J], src = CM_SRC[J], save = CM_SAVE[J], tree = CM_TREE[J];
        if constexpr (src >= 0) {
#pragma unroll
            for (int i = 0; i < 12; ++i) C.T[i] = RS ? C.S[src][i] : lds_d(L, L.slot_base + 12 * src + i);
        }
        double q = 0.0;
        if constexpr (var >= 0) q = C.q[var];
        if constexpr (kind >= SMPLX_TK_FIXED_T) apply_joint_const<J, src == SMPLX_SRC_ROOT>(q, C.T);
        else apply_joint(&M->joints[J], q, C.T, src == SMPLX_SRC_ROOT);
        if constexpr (save >= 0) {
#pragma unroll
            for (int i = 0; i < 12; ++i) { if constexpr (RS) C.S[save][i] = C.T[i]; else lds_d(L, L.slot_base + 12 * save + i) = C.T[i]; }
        }
        // the lookup issued at the previous tree has had this joint's sincos and products to land behind
        if constexpr (PT >= 0) {
#ifndef ABL_NO_TREES
            if (!resolve_root<PT>(M, L, g, C.Tp, C.pd2, lookups)) return false;
#endif
        }
        if constexpr (tree >= 0) {
            double rp[3];
#ifdef ABL_NO_TREES
            rp[0] = C.T[3]; rp[1] = C.T[7]; rp[2] = C.T[11];
#else
            issue_root<tree>(g, C, lookups, rp);
#endif
            constexpr int slot = CM_ROOT_SLOT[tree];
            if constexpr (slot >= 0) { C.roots[3 * slot] = rp[0]; C.roots[3 * slot + 1] = rp[1]; C.roots[3 * slot + 2] = rp[2]; }
#ifndef ABL_NO_PAIRS
            const_pairs<tree, CM_PAIR_FIRST[tree], CM_PAIR_FIRST[tree + 1]>(L, C, rp);
#endif
            return const_chain<J + 1, tree, RS>(M, L, g, C, lookups);
        } else {
            return const_chain<J + 1, -1, RS>(M, L, g, C, lookups);
        }
    } else {
        if constexpr (PT >= 0) {
#ifndef ABL_NO_TREES
            return resolve_root<PT>(M, L, g, C.Tp, C.pd2, lookups);
#else
            return true;
#endif
        } else {
            return true;
        }
    }
}

// planning-link chain (planning_fk below) over the on-chain joints
template <int J, bool First>
__device__ __forceinline__ void const_planning_chain(const ModelLds* __restrict__ M, const double* __restrict__ q, double T[12])
{
    if constexpr (J < CM_NJ) {
        if constexpr (CM_ON_CHAIN[J] != 0) {
            constexpr int kind = CM_KIND[J], var = CM_VAR[J];
            double qv = 0.0;
            if constexpr (var >= 0) {
                qv = q[var];
                if constexpr (CM_VAR_TYPE[var] == SMPLX_JT_CONTINUOUS) qv = smplx_normalize_angle(qv);
            }
            if constexpr (kind >= SMPLX_TK_FIXED_T) apply_joint_const<J, First>(qv, T);
            else apply_joint(&M->joints[J], qv, T, First);
            const_planning_chain<J + 1, false>(M, q, T);
        } else {
            const_planning_chain<J + 1, First>(M, q, T);
        }
    }
}
#endif   // SMPLX_CONST_MODEL

#ifdef SMPLX_CONST_MODEL
// fk_two_links for the per-robot build: the transforms of the links at joints ja and jb, the chain walked as in
// const_chain (same operations in the same order: identical bits), stopping behind the later of the two
template <int J, bool RS = false>
__device__ __forceinline__ void const_two_links(const ModelLds* __restrict__ M, const ThreadLds& L, double T[12], const double* q,
                                                int ja, int jb, int last, double Ta[12], double Tb[12], double (&S)[SMPLX_MAX_SLOTS][12])
{
    if constexpr (J < CM_NJ) {
        if (J > last) return;
        constexpr int kind = CM_KIND[J], var = CM_VAR[J], src = CM_SRC[J], save = CM_SAVE[J];
        if constexpr (src >= 0) {
#pragma unroll
            for (int i = 0; i < 12; ++i) T[i] = RS ? S[src][i] : lds_d(L, L.slot_base + 12 * src + i);
        }
        double qv = 0.0;
        if constexpr (var >= 0) qv = q[var];
        if constexpr (kind >= SMPLX_TK_FIXED_T) apply_joint_const<J, src == SMPLX_SRC_ROOT>(qv, T);
        else apply_joint(&M->joints[J], qv, T, src == SMPLX_SRC_ROOT);
        if constexpr (save >= 0) {
#pragma unroll
            for (int i = 0; i < 12; ++i) { if constexpr (RS) S[save][i] = T[i]; else lds_d(L, L.slot_base + 12 * save + i) = T[i]; }
        }
        if (J == ja) {
#pragma unroll
            for (int i = 0; i < 12; ++i) Ta[i] = T[i];
        }
        if (J == jb) {
#pragma unroll
            for (int i = 0; i < 12; ++i) Tb[i] = T[i];
        }
        const_two_links<J + 1, RS>(M, L, T, q, ja, jb, last, Ta, Tb, S);
    }
}
#endif

// sphere tree vs sphere tree (self_collision_model.cpp:1093-1218); false = collision
template <bool RS = false>
__device__ __forceinline__ bool check_pair_full(const ModelLds* __restrict__ M, const ThreadLds& L, const EdgeRef& e,
                                             int ta, int tb)
{
    double Ta[12], Tb[12];
#ifdef SMPLX_CONST_MODEL
    {
        // per-robot build: the chain as straight-line code (the generic loop reads every joint record from LDS and
        // dispatches on its kind: under random configurations, where root spheres of checked pairs overlap in most
        // waves, it was 78 % of the K2 micro-benchmark)
        double T[12], q[CM_NV];
#pragma unroll
        for (int i = 0; i < 12; ++i) { T[i] = 0.0; Ta[i] = 0.0; Tb[i] = 0.0; }
#pragma unroll
        for (int v = 0; v < CM_NV; ++v) q[v] = lds_d(L, L.q_base + v);
        const int ja = M->tree_joint[ta], jb = M->tree_joint[tb];
        double S[SMPLX_MAX_SLOTS][12];
        const_two_links<0, RS>(M, L, T, q, ja, jb, ja > jb ? ja : jb, Ta, Tb, S);
    }
#else
    fk_two_links(M, L, e, M->tree_joint[ta], M->tree_joint[tb], Ta, Tb);
#endif
    int sp = 0;
    int na = M->tree_first[ta + 1] - 1, nb = M->tree_first[tb + 1] - 1;
    while (true) {
        const LDS_AS SmplxNode& A = L.nodes[na];
        const LDS_AS SmplxNode& B = L.nodes[nb];
        double ca[3] = {A.c[0], A.c[1], A.c[2]}, cb[3] = {B.c[0], B.c[1], B.c[2]};
        double pa[3], pb[3];
        xform(Ta, ca, pa);
        xform(Tb, cb, pb);
        const double dx = pb[0] - pa[0], dy = pb[1] - pa[1], dz = pb[2] - pa[2];
        const double cd2 = (dx * dx + dy * dy) + dz * dz;
        const double rr = A.r + B.r;
        if (!(cd2 > rr * rr)) {
            const bool la = A.left < 0, lb = B.left < 0;
            if (la && lb) return false;   // leaf x leaf: the ACM lookup by sphere name never matches (:1136)
            bool split_a;
            if (la) split_a = false;
            else if (lb) split_a = true;
            else split_a = A.r > B.r;
            // both children are visited unless pruned; visiting order does not change the boolean
            if (split_a) {
                lds_b(L, sp++) = (unsigned char)A.right; lds_b(L, sp++) = (unsigned char)nb;
                na = A.left;
            } else {
                lds_b(L, sp++) = (unsigned char)na; lds_b(L, sp++) = (unsigned char)B.right;
                nb = B.left;
            }
            continue;
        }
        if (sp == 0) break;
        nb = lds_b(L, --sp);
        na = lds_b(L, --sp);
    }
    return true;
}

// CollisionSpace::isStateValid for one configuration (collision_space.cpp:532-536 ->
// self_collision_model.cpp:407-428): group trees vs grid in chain order, then the checked
// link pairs sphere-vs-sphere.
// the configuration's joint values are already staged in the thread's LDS slots (stage_config or the caller itself)
template <bool RS = false>
__device__ __forceinline__ bool config_valid_staged(const ModelLds* __restrict__ M, const ThreadLds& L, const SmplxGridDev& g,
                                                    const EdgeRef& e, int& lookups)
{
    double T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = 0.0;
    bool pair_hit = false, recheck_all = false;
    PendingPairs P;   // queued (earlier tree, later tree) pairs
    P.w0 = 0; P.w1 = 0; P.w2 = 0; P.n = 0;
#ifdef ABL_NO_FK
    lookups += (int)e.alpha; return true;
#endif
#ifdef SMPLX_CONST_MODEL
    {
        ChainState C;
#pragma unroll
        for (int i = 0; i < 12; ++i) C.T[i] = 0.0;
#pragma unroll
        for (int v = 0; v < CM_NV; ++v) C.q[v] = lds_d(L, L.q_base + v);
        C.pair_hit = false; C.recheck_all = false; C.P = P;
        C.pd2 = 0;
        if (!const_chain<0, -1, RS>(M, L, g, C, lookups)) return false;
        pair_hit = C.pair_hit; recheck_all = C.recheck_all;
        const PendingPairs filled = C.P;
        P = filled;
    }
    const int nj = 0;
#else
    const int nj = M->njoints;
#endif
    JointHead cur = load_joint_head(M, L, 0);
    for (int j = 0; j < nj; ++j) {
        // the next joint's record is requested from LDS now and consumed an iteration later, so its latency hides
        // behind this joint's sincos and products
        JointHead nxt = cur;
        if (j + 1 < nj) nxt = load_joint_head(M, L, j + 1);
        if (cur.src >= 0) {
#pragma unroll
            for (int i = 0; i < 12; ++i) T[i] = lds_d(L, L.slot_base + 12 * cur.src + i);
        }
        if (cur.kind >= SMPLX_TK_FIXED_T) apply_joint_t(cur.kind, cur.tx, cur.ty, cur.tz, cur.q, T, cur.src == SMPLX_SRC_ROOT);
        else apply_joint(&M->joints[j], cur.q, T, cur.src == SMPLX_SRC_ROOT);
        if (cur.save_slot >= 0) {
#pragma unroll
            for (int i = 0; i < 12; ++i) lds_d(L, L.slot_base + 12 * cur.save_slot + i) = T[i];
        }
        const int jtree = cur.tree;
        cur = nxt;
        if (jtree >= 0) {
            const int t = jtree;
            double rp[3];
#ifdef ABL_NO_TREES
            rp[0] = T[3]; rp[1] = T[7]; rp[2] = T[11];
#else
            if (!check_tree(M, L, g, t, T, lookups, rp)) return false;   // voxel collision: the reference stops here too
#endif
            const int slot = M->tree_root_slot[t];
            if (slot >= 0) {
                lds_d(L, L.root_base + 3 * slot + 0) = rp[0];
                lds_d(L, L.root_base + 3 * slot + 1) = rp[1];
                lds_d(L, L.root_base + 3 * slot + 2) = rp[2];
            }
            // checked link pairs whose later tree is t: root-vs-root now (self_collision_model.cpp:1111-1123);
            // anything the roots do not settle is queued and resolved after the chain (the slow path reuses the
            // transform slots).  A hit does not stop the voxel pass: the reference runs ALL voxel checks before
            // the first pair (self_collision_model.cpp:418-421), so lookup tallies stay identical.
            const LDS_AS SmplxNode& B = L.nodes[M->tree_first[t + 1] - 1];
#ifdef ABL_NO_PAIRS
            for (int k = 0; k < 0; ++k) {
#else
            for (int k = M->pair_first[t]; k < M->pair_first[t + 1]; ++k) {
#endif
                const int ta = M->pair_other[k];
                const int sa = M->tree_root_slot[ta];
                const LDS_AS SmplxNode& A = L.nodes[M->tree_first[ta + 1] - 1];
                // pairs are stored (group-earlier, group-later); the subtraction order follows that
                const bool a_first = ta < t;
                const double ax = lds_d(L, L.root_base + 3 * sa + 0), ay = lds_d(L, L.root_base + 3 * sa + 1),
                             az = lds_d(L, L.root_base + 3 * sa + 2);
                const double dx = a_first ? rp[0] - ax : ax - rp[0];
                const double dy = a_first ? rp[1] - ay : ay - rp[1];
                const double dz = a_first ? rp[2] - az : az - rp[2];
                const double cd2 = (dx * dx + dy * dy) + dz * dz;
                const double rr = a_first ? A.r + B.r : B.r + A.r;
                if (cd2 > rr * rr) continue;
                if (A.left < 0 && B.left < 0) { pair_hit = true; continue; }
                // queue (ta, t): 8 bits each, up to 4 pairs in the 64-bit word; more -> recheck everything
                if (!pend_push(P, ta, t)) recheck_all = true;
            }
        }
    }
    // unresolved pairs (normally none): one call site for the slow path, so it can be inlined without
    // putting the model view into scratch memory
    const int total = recheck_all ? M->pair_first[M->ntrees] : P.n;
    int tcur = 0;
    for (int i = 0; i < total && !pair_hit; ++i) {
        int ta, t;
        if (recheck_all) {
            while (i >= M->pair_first[tcur + 1]) ++tcur;
            t = tcur;
            ta = M->pair_other[i];
        } else {
            const int code = pend_get(P, i);
            ta = code >> 8;
            t = code & 0xFF;
        }
        const int a = ta < t ? ta : t, b = ta < t ? t : ta;
        if (!check_pair_full<RS>(M, L, e, a, b)) pair_hit = true;
    }
    return !pair_hit;
}

template <bool RS = false>
__device__ __forceinline__ bool config_valid(const ModelLds* __restrict__ M, const ThreadLds& L, const SmplxGridDev& g,
                                             const EdgeRef& e, int& lookups)
{
#ifndef ABL_NO_FK
    stage_config(M, L, e);
#endif
    return config_valid_staged<RS>(M, L, g, e, lookups);
}

// CollisionSpace::isStateToStateValid (collision_space.cpp:538-581).  first_wp = 1 skips waypoint 0
// (the start configuration), whose result the caller already has.
template <bool RS = false>
__device__ __forceinline__ bool edge_valid(const ModelLds* __restrict__ M, const ThreadLds& L, const SmplxGridDev& g,
                                           const double* __restrict__ start, const double* __restrict__ finish,
                                           bool start_known, bool start_valid, int& lookups, int& waypoints)
{
    // robot_motion_collision_model.cpp:371-407, .h:352-366, 173-181
    double motion = 0.0;
    const int nv = MV_NVARS(M);
    MV_UNROLL
    for (int v = 0; v < nv; ++v) {
        const int ty = MV_TYPE(M, v);
        const double sv = start[v], fv = finish[v];
        if (ty == SMPLX_JT_CONTINUOUS) motion += MV_K(M, v) * fabs(smplx_shortest_angle_diff(fv, sv));
        else if (ty == SMPLX_JT_REVOLUTE) motion += MV_K(M, v) * fabs(fv - sv);
        else if (ty == SMPLX_JT_PRISMATIC) motion += fabs(fv - sv);
    }
    int W = 0;
    if (motion != 0.0) {
        W = (int)ceil(motion / 0.05) + 1;
        if (W < 2) W = 2;
    }
    waypoints = W;
    if (W == 0) return true;
    if (start_known && !start_valid) return false;
    const double inv = 1.0 / (double)(W - 1);
    EdgeRef e;
    e.start = start;
    e.finish = finish;
    if (W > 5) {
        for (int i = 0; i < 5; ++i) {
            for (int j = i; j < W; j += 5) {
                if (j == 0 && start_known) continue;
                e.alpha = (double)j * inv;
                if (!config_valid<RS>(M, L, g, e, lookups)) return false;
            }
        }
    } else {
        for (int j = start_known ? 1 : 0; j < W; ++j) {
            e.alpha = (double)j * inv;
            if (!config_valid<RS>(M, L, g, e, lookups)) return false;
        }
    }
    return true;
}

// planning-link position ("KDL" FK restated as the same serial chain; kdl_robot_model.cpp:400-423,
// continuous joints normalised first :191-198)
__device__ __forceinline__ void planning_fk(const ModelLds* __restrict__ M, const double* __restrict__ q, double p[3])
{
    double T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = 0.0;
#ifdef SMPLX_CONST_MODEL
    const_planning_chain<0, true>(M, q, T);
    p[0] = T[3]; p[1] = T[7]; p[2] = T[11];
    return;
#endif
    bool first = true;
    const int nj = M->njoints;
    for (int j = 0; j < nj; ++j) {
        JointPtr jt = &M->joints[j];
        if (!jt->on_chain) continue;
        double qv = 0.0;
        if (jt->var >= 0) {
            qv = q[jt->var];
            if (MV_TYPE(M, jt->var) == SMPLX_JT_CONTINUOUS) qv = smplx_normalize_angle(qv);
        }
        apply_joint(jt, qv, T, first);
        first = false;
    }
    p[0] = T[3]; p[1] = T[7]; p[2] = T[11];
}

__device__ __forceinline__ void world_to_cell(const SmplxGridDev& g, const double p[3], int c[3])
{
    c[0] = (int)(g.inv_res * (p[0] - g.origin_minus_res[0]) + 0.5) - 1;
    c[1] = (int)(g.inv_res * (p[1] - g.origin_minus_res[1]) + 0.5) - 1;
    c[2] = (int)(g.inv_res * (p[2] - g.origin_minus_res[2]) + 0.5) - 1;
}

// BFS_3D::inBounds / getNode (bfs3d.h:151-155, 213-220)
__device__ __forceinline__ bool bfs_in_bounds(const SmplxBfsDev& b, const int c[3])
{
    return !(c[0] < 0 || c[1] < 0 || c[2] < 0 || c[0] >= b.dim_x - 2 || c[1] >= b.dim_y - 2 || c[2] >= b.dim_z - 2);
}
__device__ __forceinline__ int bfs_dist(const SmplxBfsDev& b, const int c[3])
{
    const size_t brick = ((size_t)(c[2] >> 3) * b.nby + (c[1] >> 3)) * b.nbx + (c[0] >> 3);
    const SMPLX_GLOBAL_AS int* dist = (const SMPLX_GLOBAL_AS int*)b.dist;     // (device memory, not a flat address: see grid_d2)
    const int v = dist[brick * SMPLX_BFS_REC + ((c[2] & 7) << 6) + ((c[1] & 7) << 3) + (c[0] & 7)];
    if (v == 0x7FFFFFFF) return v;
    return ((v ^ b.tag_word) & b.tag_mask) != 0 ? -1 : (v & ~b.tag_mask);     // another run's value: UNDISCOVERED
}

// BfsHeuristic::getBfsCostToGoal (bfs_heuristic.cpp:355-366)
__device__ __forceinline__ int bfs_cost_to_goal(const SmplxBfsDev& b, const int c[3])
{
    if (!bfs_in_bounds(b, c)) return 32767;
    const int d = bfs_dist(b, c);
    if (d == 0x7FFFFFFF) return 32767;
    return b.cost_per_cell * d;
}

// KDLRobotModel::checkJointLimits (kdl_robot_model.cpp:173-189, 210-235)
__device__ __forceinline__ bool check_joint_limits(const ModelLds* __restrict__ M, const double* __restrict__ q)
{
    const int nv = MV_NVARS(M);
    MV_UNROLL
    for (int v = 0; v < nv; ++v) {
        const double a_min = MV_MIN(M, v), a_max = MV_MIN_NORM(M, v);
        double a = q[v];
        if (fabs(a) > SMPLX_2PI) a = fmod(a, SMPLX_2PI);
        while (a > a_max) a -= SMPLX_2PI;
        while (a < a_min) a += SMPLX_2PI;
        if (a < MV_MIN(M, v) || a > MV_MAX(M, v)) return false;
    }
    return true;
}

// ManipLattice::stateToCoord for one variable (manip_lattice.cpp:1263-1289)
__device__ __forceinline__ int var_to_coord(const ModelLds* __restrict__ M, int v, double x)
{
    const double delta = MV_COORD_DELTA(M, v);
    const int ty = MV_TYPE(M, v);
    if (ty == SMPLX_JT_CONTINUOUS) {
        const double pos = smplx_normalize_angle_positive(x);
        int c = (int)((pos + delta * 0.5) / delta);
        if (c == MV_COORD_VALS(M, v)) c = 0;
        return c;
    }
    // bounded variables (every non-continuous variable of the plain-text model has limits)
    return (int)(((x - MV_MIN(M, v)) / delta) + 0.5);
}

// ManipLattice::getHashEntry (manip_lattice.cpp:1302-1316) against the device copy of the state table: state id of a
// discretised coordinate, -1 if the host has not committed it (yet)
// Inserts of the same launch may still be running (they ride at the head of the batch's first kernel): a slot whose tag
// is negative is being filled.  No slot between a coordinate's home and its own slot can have been empty since it was
// inserted, so meeting an empty or a busy slot first means the coordinate was not in the table before this launch.
// ConcurrentInserts: inserts may run in other workgroups of the SAME launch (k_small_batch: the extra blocks of
// table_insert_block).  In the pipeline the inserts ride with the first kernel and the lookups run in the last one, a
// kernel boundary later, where plain loads do (acquires there cost k_pipe_finish 15 -> 33 us, measured).
template <bool ConcurrentInserts>
__device__ __forceinline__ int table_lookup(const SmplxTableDev& T, const int* __restrict__ c, int nv)
{
    if (!T.slots) return -1;
    unsigned int i = smplx_coord_hash(c, nv) & T.mask;
    while (true) {
        const SMPLX_GLOBAL_AS int* sl = as_global(T.slots) + (size_t)i * T.stride;
        // ConcurrentInserts (k_small_batch): table_insert_item publishes the tag with a release after the coordinates, from
        // another workgroup; the tag is read with an acquire and the coordinates with loads that bypass this CU's L1, which is
        // never refreshed by another CU's stores -- a plain load could compare against a stale line (zeros, or half a
        // coordinate) and return another state's id
        const int tag = ConcurrentInserts ? __atomic_load_n(&sl[0], __ATOMIC_ACQUIRE) : __atomic_load_n(&sl[0], __ATOMIC_RELAXED);
        if (tag <= 0) return -1;          // free, or being filled: a miss (the host resolves misses)
        bool same = true;
        for (int v = 0; v < nv; ++v) same = same && (ConcurrentInserts ? __atomic_load_n(&sl[1 + v], __ATOMIC_RELAXED) : sl[1 + v]) == c[v];
        if (same) return tag - 1;
        i = (i + 1) & T.mask;
    }
}

// ManipLattice::createHashEntry (manip_lattice.cpp:1318-1354), device side: the host assigns ids in commit order and
// sends the (query, id, coordinate) triples of the states created since the last batch; a slot is claimed with one CAS
// on its tag and filled afterwards (lookups run in later launches of the same stream).  items: n x (nvars + 2) int32.
__device__ __forceinline__ void table_insert_item(const SmplxSpaceDev* __restrict__ S, const SmplxSpaceDev* const* __restrict__ stab,
                                                  const int* __restrict__ it, int nvars)
{
    const SmplxTableDev T = (stab ? stab[it[0]] : S)->table;
    if (!T.slots) return;
    const int id = it[1];
    int c[SMPLX_MAX_VARS];
    for (int v = 0; v < nvars; ++v) c[v] = it[2 + v];   // the items may live in pinned host memory: read them once
    unsigned int k = smplx_coord_hash(c, nvars) & T.mask;
    while (true) {
        SMPLX_GLOBAL_AS int* sl = as_global(T.slots) + (size_t)k * T.stride;
        // claim with a negative ("busy") tag, fill, publish: a concurrent lookup never sees a half-written slot as a hit
        if (atomicCAS((int*)&sl[0], 0, -(id + 1)) == 0) {
            for (int v = 0; v < nvars; ++v) __atomic_store_n(&sl[1 + v], c[v], __ATOMIC_RELAXED);
            __atomic_store_n(&sl[0], id + 1, __ATOMIC_RELEASE);
            return;
        }
        k = (k + 1) & T.mask;
    }
}

// bulk inserts (k_table_insert): every thread of the launch takes its share
__device__ __forceinline__ void table_insert_items(const SmplxSpaceDev* __restrict__ S, const SmplxSpaceDev* const* __restrict__ stab,
                                                   const int* __restrict__ items, int n, int nvars)
{
    const int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) table_insert_item(S, stab, items + (size_t)i * (nvars + 2), nvars);
}

// The inserts that ride with a batch's first kernel take EXTRA blocks behind the `first_block` working ones, so they run
// beside the batch instead of in front of it (a lookup that misses one of them just reports "unknown").  Returns true
// for such a block: the caller returns at once.
__device__ __forceinline__ bool table_insert_block(const SmplxSpaceDev* __restrict__ S, const SmplxSpaceDev* const* __restrict__ stab,
                                                   const int* __restrict__ items, int n, int first_block)
{
    if ((int)blockIdx.x < first_block) return false;
    const int i = ((int)blockIdx.x - first_block) * (int)blockDim.x + (int)threadIdx.x;
    const int nvars = S->model.nvars;
    if (i < n) table_insert_item(S, stab, items + (size_t)i * (nvars + 2), nvars);
    return true;
}

extern "C" __global__ void __launch_bounds__(BLOCK)
k_table_insert(const SmplxSpaceDev* __restrict__ S, const SmplxSpaceDev* const* __restrict__ stab, const int* __restrict__ items,
               int n, int nvars)
{
    table_insert_items(S, stab, items, n, nvars);
}

// Cooperative copy of the packed model (a few KB) into LDS in 16-byte pieces, all loads of a thread issued before
// its first store; every later read of the model is a uniform-address LDS broadcast instead of a dependent
// global load.  Returns the view.
__device__ __forceinline__ ModelLds stage_model(const SmplxSpaceDev* __restrict__ S, unsigned char* smem, int nthreads = BLOCK)
{
    typedef double __attribute__((ext_vector_type(2))) d2_t;
    const int* hdr = reinterpret_cast<const int*>(S->model_blob);
    const d2_t* src = reinterpret_cast<const d2_t*>(S->model_blob);
    d2_t* dst = reinterpret_cast<d2_t*>(smem);
    const int total = hdr[SMPLX_BH_BYTES] / 16;
    d2_t v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = threadIdx.x + k * nthreads;
        if (i < total) v[k] = src[i];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = threadIdx.x + k * nthreads;
        if (i < total) dst[i] = v[k];
    }
    for (int i = threadIdx.x + 4 * nthreads; i < total; i += nthreads) dst[i] = src[i];
    ModelLds M;
    M.njoints = hdr[SMPLX_BH_NJOINTS]; M.nvars = hdr[SMPLX_BH_NVARS]; M.ntrees = hdr[SMPLX_BH_NTREES];
    M.nnodes = hdr[SMPLX_BH_NNODES]; M.npairs = hdr[SMPLX_BH_NPAIRS]; M.nslots = hdr[SMPLX_BH_NSLOTS];
    M.nroot = hdr[SMPLX_BH_NROOT];
    LDS_AS unsigned char* base = (LDS_AS unsigned char*)smem;
    M.joints = (JointPtr)(base + hdr[SMPLX_BH_OFF_JOINTS]);
    M.nodes = (NodePtr)(base + hdr[SMPLX_BH_OFF_NODES]);
    IntPtr ip = (IntPtr)(base + hdr[SMPLX_BH_OFF_INTS]);
    M.tree_first = ip; ip += M.ntrees + 1;
    M.tree_joint = ip; ip += M.ntrees;
    M.tree_root_slot = ip; ip += M.ntrees;
    M.pair_first = ip; ip += M.ntrees + 1;
    M.pair_other = ip;
    DblPtr dp = (DblPtr)(base + hdr[SMPLX_BH_OFF_VARD]);
    M.var_min = dp; M.var_max = dp + M.nvars; M.var_min_norm = dp + 2 * M.nvars; M.var_k = dp + 3 * M.nvars;
    M.coord_delta = dp + 4 * M.nvars;
    IntPtr vp = (IntPtr)(base + hdr[SMPLX_BH_OFF_VARI]);
    M.coord_vals = vp; M.var_type = vp + M.nvars;
    return M;
}

// model + per-thread scratch (root-position slots, saved transforms, DFS stack)
__device__ __forceinline__ ThreadLds setup_lds(const SmplxSpaceDev* __restrict__ S, unsigned char* smem, ModelLds* Mv,
                                               int nthreads = BLOCK, bool slots_in_lds = true)
{
    ThreadLds L;
    *Mv = stage_model(S, smem, nthreads);
    L.stride = nthreads;
    const int* hdr = reinterpret_cast<const int*>(S->model_blob);
    L.nodes = Mv->nodes;
    L.d = (LDS_AS double*)((LDS_AS unsigned char*)smem + hdr[SMPLX_BH_BYTES]);
#ifdef SMPLX_CONST_MODEL
    const int nroot = 0;   // per-robot build: the root positions that lead a checked pair live in registers (ChainState::roots)
#else
    const int nroot = Mv->nroot;
#endif
    L.root_base = 0;
    L.slot_base = 3 * nroot;
    const int nslots = slots_in_lds ? Mv->nslots : 0;      // (a kernel that keeps the saved transforms in registers: const_chain<.., true>)
    L.q_base = 3 * nroot + 12 * nslots;
    const int nd = 3 * nroot + 12 * nslots + Mv->nvars;
    L.stk = (LDS_AS unsigned char*)(L.d + nd * nthreads);
    __syncthreads();
    return L;
}

// kernels that only need the model (no per-thread scratch)
__device__ __forceinline__ ModelLds setup_model_only(const SmplxSpaceDev* __restrict__ S, unsigned char* smem)
{
#if defined(SMPLX_CONST_MODEL) && !CM_NEEDS_JOINTS
    // per-robot build: the planning-link chain and the per-variable data are literals, nothing is read from LDS
    ModelLds M = {};
    M.njoints = CM_NJ; M.nvars = CM_NV; M.ntrees = CM_NT;
    return M;
#else
    ModelLds M = stage_model(S, smem);
    __syncthreads();
    return M;
#endif
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------

extern "C" __global__ void __launch_bounds__(BLOCK, 2)   // >= 2 waves per SIMD: at most 256 VGPRs, whichever compiler builds it
k_state_prep(const SmplxSpaceDev* __restrict__ S, const double* __restrict__ Q, const int64_t* __restrict__ refs, int B,
             double* __restrict__ goal_dist, unsigned char* __restrict__ parent_valid, int* __restrict__ parent_lookups,
        const SmplxSpaceDev* const* __restrict__ stab, const unsigned short* __restrict__ state_q)
{
    extern __shared__ __align__(16) unsigned char smem[];
    ModelLds Mv;
    ThreadLds L = setup_lds(S, smem, &Mv);
    const ModelLds* M = &Mv;
    const SmplxGridDev grid = S->grid;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= B) return;
    const SmplxBfsDev bfs = (stab ? stab[state_q[i]] : S)->bfs;   // per-query data in a cross-query batch
    const double* q = Q + (refs ? refs[i] : (int64_t)i) * MV_NVARS(M);
    double p[3];
    planning_fk(M, q, p);
    // BfsHeuristic::getMetricGoalDistance (bfs_heuristic.cpp:129-138)
    int c[3];
    world_to_cell(grid, p, c);
    double gd;
    if (!bfs_in_bounds(bfs, c)) gd = (double)0x7FFFFFFF * grid.res;
    else gd = (double)bfs_dist(bfs, c) * grid.res;
    goal_dist[i] = gd;
    EdgeRef e;
    e.start = q; e.finish = q; e.alpha = 0.0;
    int lk = 0;
    const bool ok = config_valid(M, L, grid, e, lk);
    parent_valid[i] = ok ? 1 : 0;
    parent_lookups[i] = lk;
}

// per-block tallies without atomics: block b owns counters[4*b .. 4*b+3] (launches on one stream serialise,
// so a plain read-modify-write is safe); the host sums the blocks (smplx_counters_read)
__device__ __forceinline__ void tally_block(unsigned long long* __restrict__ counters, int ev, int va, int lk, int pf,
                                            int cfgs, int slk)
{
    __shared__ int t_acc[BLOCK / 64][SMPLX_TALLIES];
    const int wv = threadIdx.x >> 6;
    for (int off = 32; off > 0; off >>= 1) {
        lk += __shfl_down(lk, off); pf += __shfl_down(pf, off); cfgs += __shfl_down(cfgs, off); slk += __shfl_down(slk, off);
    }
    if ((threadIdx.x & 63) == 0) {
        t_acc[wv][0] = ev; t_acc[wv][1] = va; t_acc[wv][2] = lk; t_acc[wv][3] = pf; t_acc[wv][4] = cfgs; t_acc[wv][5] = slk;
    }
    __syncthreads();
    if (threadIdx.x < SMPLX_TALLIES) {
        int v = 0;
#pragma unroll
        for (int k = 0; k < BLOCK / 64; ++k) v += t_acc[k][threadIdx.x];
        counters[(size_t)blockIdx.x * SMPLX_TALLIES + threadIdx.x] += (unsigned long long)v;
    }
}

// manip_lattice_action_space.cpp:662-691
__device__ __forceinline__ bool mprim_active(const SmplxActionsDev& A, double goal_dist, int type)
{
    if (type == SMPLX_MP_LONG) {
        if (A.use_long_and_short) return true;
        const bool near_goal = goal_dist <= A.thresh[SMPLX_MP_SHORT];
        return !(A.enabled[SMPLX_MP_SHORT] && near_goal);
    } else if (type == SMPLX_MP_SHORT) {
        if (A.use_long_and_short) return A.enabled[type] != 0;
        const bool near_goal = goal_dist <= A.thresh[type];
        return A.enabled[type] && near_goal;
    }
    return A.enabled[type] && goal_dist <= A.thresh[type];
}

// One (state, primitive) pair through the whole GetSuccs loop body in ONE thread (manip_lattice.cpp:1471-1535):
// gating, successor joint values, limits, the edge's waypoints in the reference's order, discretisation, goal test,
// heuristic.  parent_ok / parent_lk: result of the state's own check (waypoint 0 of every edge).
struct EdgeTally { int flags, lookups, performed, evaluated; };

__device__ __forceinline__ EdgeTally expand_edge(const ModelLds* __restrict__ M, const ThreadLds& L, const SmplxSpaceDev* __restrict__ S,
                                                 const SmplxSpaceDev* __restrict__ Sq, const SmplxGridDev& grid,
                                                 const double* __restrict__ Q, const int64_t* __restrict__ refs, long long tid,
                                                 const double* __restrict__ goal_dist, bool parent_ok, int parent_lk,
                                                 unsigned char* __restrict__ out_flags, int* __restrict__ out_coord,
                                                 double* __restrict__ out_q, int* __restrict__ out_h, int* __restrict__ out_cost,
                                                 int* __restrict__ out_lookups)
{
    const SmplxActionsDev& A = S->actions;
    const int nprims = A.nprims;
    int flags = SMPLX_F_INACTIVE;
    int lookups = 0;
    int performed = 0;   // lookups this thread itself issued (waypoints >= 1)
    int evaluated = 0;
    {
        const int si = (int)(tid / nprims);
        const int pi = (int)(tid - (long long)si * nprims);
        const int nv = MV_NVARS(M);
        const double* parent = Q + (refs ? refs[si] : (int64_t)si) * nv;
        double* sq = out_q + tid * nv;
        int* sc = out_coord + tid * nv;
        const SmplxBfsDev bfs = Sq->bfs;
        const int type = A.type[pi];
        int h = 0, cost = 0;
        bool have_action = false;
        if (mprim_active(A, goal_dist[si], type)) {
            if (type == SMPLX_MP_LONG || type == SMPLX_MP_SHORT) {
                // applyMotionPrimitive (manip_lattice_action_space.cpp:575-621)
                double d0 = A.delta[pi][0], d1 = nv > 1 ? A.delta[pi][1] : 0.0;
                if (A.xy_rotate_by_var3 && nv > 3) {
                    double s, c;
                    smplx_sincos(parent[3], &s, &c);
                    const double a0 = d0, a1 = d1;
                    d0 = c * a0 + (-s) * a1;
                    d1 = s * a0 + c * a1;
                }
                MV_UNROLL
                for (int v = 0; v < nv; ++v) {
                    const double d = v == 0 ? d0 : (v == 1 ? d1 : A.delta[pi][v]);
                    sq[v] = d + parent[v];
                }
                have_action = true;
            } else if (type == SMPLX_MP_SNAP_XYZ_RPY && Sq->goal.type == SMPLX_GOAL_JOINT) {
                MV_UNROLL
                for (int v = 0; v < nv; ++v) sq[v] = Sq->goal.angles[v];   // :551-559
                have_action = true;
            }
        }
        if (have_action) {
            evaluated = 1;
            flags = 0;
            if (!check_joint_limits(M, sq)) {
                flags = SMPLX_F_LIMITS;
            } else {
                int W = 0;
                int lk = 0;
                const bool ok = edge_valid(M, L, grid, parent, sq, true, parent_ok, lk, W);
                lookups = lk;
                performed = lk;
                if (W > 0) lookups += parent_lk;   // waypoint 0, done once per state
                if (!ok) {
                    flags = SMPLX_F_COLLISION;
                } else {
                    MV_UNROLL
                    for (int v = 0; v < nv; ++v) sc[v] = var_to_coord(M, v, sq[v]);
                    bool is_goal;
                    double p[3];
                    planning_fk(M, sq, p);
                    if (Sq->goal.type == SMPLX_GOAL_JOINT) {      // manip_lattice.cpp:1596-1606
                        is_goal = true;
                        MV_UNROLL
                        for (int v = 0; v < nv; ++v)
                            if (fabs((double)(sc[v] - Sq->goal.coord[v])) > Sq->goal.angle_tol[v]) is_goal = false;
                    } else {                                      // XYZ goal :1672-1687
                        is_goal = fabs(p[0] - Sq->goal.xyz[0]) <= Sq->goal.xyz_tol[0] &&
                                  fabs(p[1] - Sq->goal.xyz[1]) <= Sq->goal.xyz_tol[1] &&
                                  fabs(p[2] - Sq->goal.xyz[2]) <= Sq->goal.xyz_tol[2];
                    }
                    int c[3];
                    world_to_cell(grid, p, c);
                    h = bfs_cost_to_goal(bfs, c);
                    cost = A.cost[pi];
                    flags = SMPLX_F_VALID | (is_goal ? SMPLX_F_GOAL : 0);
                }
            }
        }
        out_flags[tid] = (unsigned char)flags;
        out_h[tid] = h;
        out_cost[tid] = cost;
        out_lookups[tid] = lookups;
    }
    EdgeTally t;
    t.flags = flags; t.lookups = lookups; t.performed = performed; t.evaluated = evaluated;
    return t;
}

extern "C" __global__ void __launch_bounds__(BLOCK, 2)   // >= 2 waves per SIMD: at most 256 VGPRs, whichever compiler builds it
k_expand(const SmplxSpaceDev* __restrict__ S, const double* __restrict__ Q, const int64_t* __restrict__ refs, int B,
         const double* __restrict__ goal_dist, const unsigned char* __restrict__ parent_valid,
         const int* __restrict__ parent_lookups,
         unsigned char* __restrict__ out_flags, int* __restrict__ out_coord, double* __restrict__ out_q,
         int* __restrict__ out_h, int* __restrict__ out_cost, int* __restrict__ out_lookups,
         unsigned long long* __restrict__ counters, const int* __restrict__ deferred_count,
        const SmplxSpaceDev* const* __restrict__ stab, const unsigned short* __restrict__ state_q)
{
    extern __shared__ __align__(16) unsigned char smem[];
    // second pass after the pipeline / the small-batch kernel (deferred_count != nullptr): nothing to do in the
    // common case.  deferred_count[0] < 0 means "no counter kept": the block looks at its own flags instead.
    const bool only_deferred = deferred_count != nullptr;
    if (only_deferred) {
        const int cnt = deferred_count[0];
        if (cnt == 0) return;
        if (cnt < 0) {
            const long long tid0 = (long long)blockIdx.x * BLOCK + threadIdx.x;
            const int mine = (tid0 < (long long)B * S->actions.nprims) ? (out_flags[tid0] & SMPLX_F_DEFERRED) : 0;
            if (!__syncthreads_or(mine)) return;
        }
    }
    ModelLds Mv;
    const SmplxActionsDev& A = S->actions;
    ThreadLds L = setup_lds(S, smem, &Mv);
    const ModelLds* M = &Mv;
    const SmplxGridDev grid = S->grid;
    const int nprims = A.nprims;
    const long long tid = (long long)blockIdx.x * BLOCK + threadIdx.x;
    bool in_range = tid < (long long)B * nprims;
    // second pass after the pipeline: only the edges it deferred (they did not fit the work list)
    if (only_deferred && in_range && !(out_flags[tid] & SMPLX_F_DEFERRED)) in_range = false;
    int flags = SMPLX_F_INACTIVE;
    int lookups = 0;
    int performed = 0;   // lookups this kernel itself issued (waypoints >= 1)
    int evaluated = 0;
    if (in_range) {
        const int si = (int)(tid / nprims);
        const SmplxSpaceDev* Sq = stab ? stab[state_q[si]] : S;   // per-query goal and BFS grid
        // fused mode: parent_valid holds 1 = valid (k_state_prep); deferred pass: the pipeline's state_bad (1 = bad)
        const bool pv = only_deferred ? parent_valid[si] == 0 : parent_valid[si] != 0;
        const EdgeTally t = expand_edge(M, L, S, Sq, grid, Q, refs, tid, goal_dist, pv, parent_lookups[si], out_flags, out_coord,
                                        out_q, out_h, out_cost, out_lookups);
        flags = t.flags; lookups = t.lookups; performed = t.performed; evaluated = t.evaluated;
    }
    // per-wave tallies: ballots instead of one atomic per lane
    if (counters) {
        const unsigned long long m_eval = __ballot(evaluated);
        const unsigned long long m_valid = __ballot((flags & SMPLX_F_VALID) != 0);
        tally_block(counters, __popcll(m_eval), __popcll(m_valid), lookups, performed, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------
// Waypoint-parallel pipeline (default).  The fused k_expand above walks an edge's waypoints one
// after another inside one thread, which leaves the chip idle at B = 4096 (about 1.6 waves per
// SIMD, each a serial fp64 chain).  The pipeline spreads the same work over (edge, waypoint) items:
//   k_pipe_prep    per state: planning-link FK -> metric goal distance
//   k_pipe_setup   per (state, primitive): gating, successor joint values, limits, waypoint count;
//                  claims a range of the work list with one atomic per wave (ballot + prefix count)
//   k_pipe_configs per work item: one configuration against the grid and the link pairs;
//                  items [0, B) are the states themselves (waypoint 0 of every edge)
//   k_pipe_finish  per (state, primitive): verdict, discretisation, goal test, heuristic, cost
// Booleans, coordinates, heuristics and costs are identical to k_expand.  Without the serial
// early exit a colliding edge has all its waypoints examined, so the lookup tally of an INVALID
// edge can exceed the reference's; for valid edges it is identical.
// ---------------------------------------------------------------------------------------------

// work item (64 bits): edge index | waypoint << 32 | waypoint count << 48, so that a configuration thread needs no
// further load to know where it sits on its edge
#define SMPLX_WP_MAX 0xFFFF
#define SMPLX_WORK_BLANK 0xFFFFFFFFFFFFFFFFull
#define SMPLX_WORK_SHARDS 8
#define SMPLX_SHARD_STRIDE 32   // ints: one 128-byte line per shard counter

extern "C" __global__ void __launch_bounds__(BLOCK)
k_pipe_prep(const SmplxSpaceDev* __restrict__ S, const double* __restrict__ Q, const int64_t* __restrict__ refs, int B,
            double* __restrict__ goal_dist, int* __restrict__ work_count,
        const SmplxSpaceDev* const* __restrict__ stab, const unsigned short* __restrict__ state_q, int* __restrict__ cmp_totals,
            const int* __restrict__ ins_items, int n_ins)
{
    extern __shared__ __align__(16) unsigned char smem[];
    // K5: the states the host committed since the last batch join the device table (createHashEntry) in extra blocks
    if (n_ins > 0 && table_insert_block(S, stab, ins_items, n_ins, (B + BLOCK - 1) / BLOCK)) return;
    const ModelLds Mv = setup_model_only(S, smem);
    const ModelLds* M = &Mv;
    const SmplxGridDev grid = S->grid;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i <= SMPLX_WORK_SHARDS) work_count[i * SMPLX_SHARD_STRIDE] = 0;   // shard counters + deferred count
    if (cmp_totals && blockIdx.x == 0)                                   // compaction counters of k_pipe_finish
        for (int k = threadIdx.x; k < SMPLX_CMP_TOTALS; k += BLOCK) cmp_totals[k] = 0;
    if (i >= B) return;
    const SmplxBfsDev bfs = (stab ? stab[state_q[i]] : S)->bfs;
    const double* q = Q + (refs ? refs[i] : (int64_t)i) * MV_NVARS(M);
    double p[3];
    planning_fk(M, q, p);
    int c[3];
    world_to_cell(grid, p, c);
    goal_dist[i] = !bfs_in_bounds(bfs, c) ? (double)0x7FFFFFFF * grid.res : (double)bfs_dist(bfs, c) * grid.res;
}

extern "C" __global__ void __launch_bounds__(BLOCK)
k_pipe_setup(const SmplxSpaceDev* __restrict__ S, const double* __restrict__ Q, const int64_t* __restrict__ refs, int B,
             const double* __restrict__ goal_dist, unsigned char* __restrict__ out_flags, double* __restrict__ out_q,
             int* __restrict__ edge_w, int* __restrict__ edge_lookups, unsigned char* __restrict__ edge_bad,
             int* __restrict__ state_lookups, unsigned char* __restrict__ state_bad,
             unsigned long long* __restrict__ work, int* __restrict__ work_count, int capacity,
        const SmplxSpaceDev* const* __restrict__ stab, const unsigned short* __restrict__ state_q)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const ModelLds Mv = setup_model_only(S, smem);
    const ModelLds* M = &Mv;
    const SmplxActionsDev& A = S->actions;
    const int nprims = A.nprims;
    const long long tid = (long long)blockIdx.x * BLOCK + threadIdx.x;
    const bool in_range = tid < (long long)B * nprims;
    int items = 0;
    int W = 0;
    int flags = SMPLX_F_INACTIVE;
    if (in_range) {
        const int si = (int)(tid / nprims);
        const int pi = (int)(tid - (long long)si * nprims);
        const int nv = MV_NVARS(M);
        const double* parent = Q + (refs ? refs[si] : (int64_t)si) * nv;
        double* sq = out_q + tid * nv;
        const SmplxSpaceDev* Sq = stab ? stab[state_q[si]] : S;
        const int type = A.type[pi];
        bool have_action = false;
        if (pi == 0) { state_lookups[si] = 0; state_bad[si] = 0; }
        if (mprim_active(A, goal_dist[si], type)) {
            if (type == SMPLX_MP_LONG || type == SMPLX_MP_SHORT) {
                double d0 = A.delta[pi][0], d1 = nv > 1 ? A.delta[pi][1] : 0.0;
                if (A.xy_rotate_by_var3 && nv > 3) {
                    double s, c;
                    smplx_sincos(parent[3], &s, &c);
                    const double a0 = d0, a1 = d1;
                    d0 = c * a0 + (-s) * a1;
                    d1 = s * a0 + c * a1;
                }
                MV_UNROLL
                for (int v = 0; v < nv; ++v) {
                    const double d = v == 0 ? d0 : (v == 1 ? d1 : A.delta[pi][v]);
                    sq[v] = d + parent[v];
                }
                have_action = true;
            } else if (type == SMPLX_MP_SNAP_XYZ_RPY && Sq->goal.type == SMPLX_GOAL_JOINT) {
                MV_UNROLL
                for (int v = 0; v < nv; ++v) sq[v] = Sq->goal.angles[v];
                have_action = true;
            }
        }
        if (have_action) {
            flags = 0;
            if (!check_joint_limits(M, sq)) {
                flags = SMPLX_F_LIMITS;
            } else {
                double motion = 0.0;
                MV_UNROLL
                for (int v = 0; v < nv; ++v) {
                    const int ty = MV_TYPE(M, v);
                    const double sv = parent[v], fv = sq[v];
                    if (ty == SMPLX_JT_CONTINUOUS) motion += MV_K(M, v) * fabs(smplx_shortest_angle_diff(fv, sv));
                    else if (ty == SMPLX_JT_REVOLUTE) motion += MV_K(M, v) * fabs(fv - sv);
                    else if (ty == SMPLX_JT_PRISMATIC) motion += fabs(fv - sv);
                }
                if (motion != 0.0) {
                    W = (int)ceil(motion / 0.05) + 1;
                    if (W < 2) W = 2;
                }
                items = W > 0 ? W - 1 : 0;
            }
        }
        edge_lookups[tid] = 0;
        edge_bad[tid] = 0;
    }
    // claim a contiguous range of the work list.  Same-address atomics serialise at ~12 ns each on this
    // chip, so: wave prefix count (shuffles) -> block total through LDS -> ONE atomic per block, spread over
    // SMPLX_WORK_SHARDS counters that live on separate 128-byte lines.
    __shared__ int wave_sum[BLOCK / 64];
    __shared__ int block_base;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int incl = items;
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wave_sum[wv] = incl;
    __syncthreads();
    const int shard = blockIdx.x % SMPLX_WORK_SHARDS;
    const int shard_cap = capacity / SMPLX_WORK_SHARDS;
    if (threadIdx.x == 0) {
        int tot = 0;
#pragma unroll
        for (int k = 0; k < BLOCK / 64; ++k) tot += wave_sum[k];
        block_base = tot > 0 ? atomicAdd(&work_count[shard * SMPLX_SHARD_STRIDE], tot) : 0;
    }
    __syncthreads();
    int first = block_base + incl - items;
    for (int k = 0; k < wv; ++k) first += wave_sum[k];
    if (in_range) {
        if (items > 0) {
            unsigned long long* wl = work + (size_t)shard * shard_cap;
            if (first + items <= shard_cap && W <= SMPLX_WP_MAX) {
                const unsigned long long base = (unsigned long long)tid | ((unsigned long long)W << 48);
                for (int k = 0; k < items; ++k) wl[first + k] = base | ((unsigned long long)(k + 1) << 32);
            } else {
                // does not fit: deferred to a fused pass (k_expand, SMPLX_F_DEFERRED); blank the part of the claim
                // that lies below the shard's capacity
                for (int k = first; k < first + items && k < shard_cap; ++k) wl[k] = SMPLX_WORK_BLANK;
                flags = SMPLX_F_DEFERRED;
                atomicAdd(&work_count[SMPLX_WORK_SHARDS * SMPLX_SHARD_STRIDE], 1);
            }
        }
        edge_w[tid] = W;
        out_flags[tid] = (unsigned char)flags;
    }
}

extern "C" __global__ void __launch_bounds__(BLOCK, 2)   // >= 2 waves per SIMD: at most 256 VGPRs, whichever compiler builds it
k_pipe_configs(const SmplxSpaceDev* __restrict__ S, const double* __restrict__ Q, const int64_t* __restrict__ refs, int B,
               const double* __restrict__ out_q, const int* __restrict__ edge_w, int* __restrict__ edge_lookups,
               unsigned char* __restrict__ edge_bad, int* __restrict__ state_lookups, unsigned char* __restrict__ state_bad,
               const unsigned long long* __restrict__ work, const int* __restrict__ work_count, int capacity)
{
    extern __shared__ __align__(16) unsigned char smem[];
#ifdef SMPLX_CONST_MODEL
    constexpr bool RS = true;      // saved link transforms in registers (as k_state_valid): LDS per block without the slots, which
                                   // is what several batches in flight, or one large one, share a CU by
#else
    constexpr bool RS = false;
#endif
    const int shard_cap = capacity / SMPLX_WORK_SHARDS;
    int pre[SMPLX_WORK_SHARDS + 1];   // prefix of the shard fill counts (claims beyond a shard's capacity were never written)
    pre[0] = 0;
#pragma unroll
    for (int k = 0; k < SMPLX_WORK_SHARDS; ++k) {
        int c = work_count[k * SMPLX_SHARD_STRIDE];
        if (c > shard_cap) c = shard_cap;
        pre[k + 1] = pre[k] + c;
    }
    const long long total = (long long)B + pre[SMPLX_WORK_SHARDS];
    if ((long long)blockIdx.x * BLOCK >= total) return;   // whole block idle: skip staging the model
#ifdef SMPLX_CONST_MODEL
    // Per-robot build: the launch covers every item (engine.hip sizes the grid for B + 3 B M items and k_pipe_setup never
    // lists more), one item per thread.  A block's life is a chain of dependent memory round trips of ~1 us each --
    // counts, model header, model bytes, work item, joint values -- in front of ~10 us of work: the item and the joint
    // values of its edge are fetched BEFORE the model is staged, so that they travel together with the model bytes
    // (5 round trips -> 3).
    if (total <= (long long)gridDim.x * BLOCK) {
        const int nprims = S->actions.nprims;
        constexpr int nv = CM_NV;
        const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
        unsigned long long it = SMPLX_WORK_BLANK;
        if (i >= B && i < total) {
            const int li = (int)(i - B);
            int sh = 0;
#pragma unroll
            for (int k = 1; k < SMPLX_WORK_SHARDS; ++k) sh += (li >= pre[k]) ? 1 : 0;
            it = work[(size_t)sh * shard_cap + (li - pre[sh])];
        }
        const bool is_state = i < B, is_item = it != SMPLX_WORK_BLANK;
        const long long edge = (long long)(it & 0xFFFFFFFFull);
        const int wp = (int)((it >> 32) & 0xFFFF);
        const int W = (int)(it >> 48);
        double qs[CM_NV], qf[CM_NV];
        if (is_state || is_item) {
            const long long si = is_state ? i : edge / nprims;
            const double* ps = Q + (refs ? refs[si] : (int64_t)si) * nv;
            const double* pf = is_state ? ps : out_q + edge * nv;
#pragma unroll
            for (int v = 0; v < nv; ++v) { qs[v] = ps[v]; qf[v] = pf[v]; }
        }
        ModelLds Mv;
        ThreadLds L = setup_lds(S, smem, &Mv, BLOCK, !RS);
        const ModelLds* M = &Mv;
        const SmplxGridDev grid = S->grid;
        if (!(is_state || is_item)) return;
        EdgeRef e;
        e.start = nullptr; e.finish = nullptr;   // config_valid_staged never dereferences them
        e.alpha = is_state ? 0.0 : (double)wp * (1.0 / (double)(W - 1));
        int lk = 0;
#ifndef ABL_NO_FK
#pragma unroll
        for (int v = 0; v < nv; ++v) {   // stage_config
            const double sv = qs[v];
            double q = sv;
            if (e.alpha != 0.0) q = sv + e.alpha * edge_diff(M, v, sv, qf[v]);
            lds_d(L, L.q_base + v) = q;
        }
#endif
        const bool ok = config_valid_staged<RS>(M, L, grid, e, lk);
        if (is_state) {
            state_lookups[i] = lk;
            if (!ok) state_bad[i] = 1;
        } else {
            atomicAdd(&edge_lookups[edge], lk);
            if (!ok) edge_bad[edge] = 1;
        }
        return;
    }
#endif
    ModelLds Mv;
    ThreadLds L = setup_lds(S, smem, &Mv, BLOCK, !RS);
    const ModelLds* M = &Mv;
    const SmplxGridDev grid = S->grid;
    const int nprims = S->actions.nprims;
    const int nv = MV_NVARS(M);
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (long long)gridDim.x * BLOCK) {
        EdgeRef e;
        int lk = 0;
        if (i < B) {   // the state itself: waypoint 0 of each of its edges
            e.start = Q + (refs ? refs[i] : i) * nv;
            e.finish = e.start;
            e.alpha = 0.0;
            const bool ok = config_valid<RS>(M, L, grid, e, lk);
            state_lookups[i] = lk;
            if (!ok) state_bad[i] = 1;
        } else {
            const int li = (int)(i - B);
            int sh = 0;
#pragma unroll
            for (int k = 1; k < SMPLX_WORK_SHARDS; ++k) sh += (li >= pre[k]) ? 1 : 0;
            const unsigned long long it = work[(size_t)sh * shard_cap + (li - pre[sh])];
            if (it == SMPLX_WORK_BLANK) continue;
            const long long edge = (long long)(it & 0xFFFFFFFFull);
            const int wp = (int)((it >> 32) & 0xFFFF);
            const int W = (int)(it >> 48);
            const int si = (int)(edge / nprims);
            e.start = Q + (refs ? refs[si] : (int64_t)si) * nv;
            e.finish = out_q + edge * nv;
            e.alpha = (double)wp * (1.0 / (double)(W - 1));
            const bool ok = config_valid<RS>(M, L, grid, e, lk);
            atomicAdd(&edge_lookups[edge], lk);
            if (!ok) edge_bad[edge] = 1;
        }
    }
}

extern "C" __global__ void __launch_bounds__(BLOCK, 2)   // holds the whole-edge walk for overflowed edges: keep it at 2 waves per SIMD
k_pipe_finish(const SmplxSpaceDev* __restrict__ S, const double* __restrict__ Q, const int64_t* __restrict__ refs, int B,
              const int* __restrict__ edge_w, const int* __restrict__ edge_lookups, const unsigned char* __restrict__ edge_bad,
              const int* __restrict__ state_lookups, const unsigned char* __restrict__ state_bad,
              unsigned char* __restrict__ out_flags, int* __restrict__ out_coord, double* __restrict__ out_q,
              int* __restrict__ out_h, int* __restrict__ out_cost, int* __restrict__ out_lookups,
              unsigned long long* __restrict__ counters, const double* __restrict__ goal_dist,
        const SmplxSpaceDev* const* __restrict__ stab, const unsigned short* __restrict__ state_q,
              int* __restrict__ out_id, SmplxCompactDev cmp)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const SmplxActionsDev& A = S->actions;
    const SmplxGridDev grid = S->grid;
    const int nprims = A.nprims;
    const long long tid = (long long)blockIdx.x * BLOCK + threadIdx.x;
    const bool in_range = tid < (long long)B * nprims;
    int flags = SMPLX_F_INACTIVE, lookups = 0, performed = 0, evaluated = 0;
    int succ_id = -1, succ_h = 0;   // K5: id of the successor's coordinate in the device state table
    int early_id = -1;
    bool have_early = false;
    int ncfg = 0, slk = 0;   // configurations k_pipe_configs checked for this edge / lookups of the state's own check
    if (in_range) flags = out_flags[tid];
    // the model and the per-thread scratch are only needed by edges that overflowed the work list (normally none)
    ModelLds Mv;
    ThreadLds L;
#if defined(SMPLX_CONST_MODEL) && !CM_NEEDS_JOINTS
    if (__syncthreads_or(flags & SMPLX_F_DEFERRED)) {
        L = setup_lds(S, smem, &Mv);
    } else {
        Mv = setup_model_only(S, smem);
        L = ThreadLds();
    }
#else
    L = setup_lds(S, smem, &Mv);
#endif
    const ModelLds* M = &Mv;
    if (in_range) {
        const int si = (int)(tid / nprims);
        const int pi = (int)(tid - (long long)si * nprims);
        const int nv = MV_NVARS(M);
        const SmplxSpaceDev* Sq = stab ? stab[state_q[si]] : S;
        const SmplxBfsDev bfs = Sq->bfs;
        if (pi == 0) { slk = state_lookups[si]; ncfg = 1; }
        int h = 0, cost = 0;
        bool deferred = false;
        if (flags & SMPLX_F_DEFERRED) {
            // the edge's waypoints did not fit the work list (normally none do not): this thread walks the whole edge
            const EdgeTally t = expand_edge(M, L, S, Sq, grid, Q, refs, tid, goal_dist, state_bad[si] == 0, state_lookups[si],
                                            out_flags, out_coord, out_q, out_h, out_cost, out_lookups);
            flags = t.flags; lookups = t.lookups; performed = t.performed; evaluated = t.evaluated;
            deferred = true;
        }
        if (!deferred && !(flags & SMPLX_F_INACTIVE)) evaluated = 1;
        if (!deferred && flags == 0) {
            const int W = edge_w[tid];
            if (W > 0) ncfg += W - 1;
            performed = edge_lookups[tid];
            const bool ok = (W == 0) || (state_bad[si] == 0 && edge_bad[tid] == 0);
            lookups = performed + (W > 0 ? state_lookups[si] : 0);
            if (!ok) {
                flags = SMPLX_F_COLLISION;
            } else {
                const double* sq = out_q + tid * nv;
                int* sc = out_coord + tid * nv;
                MV_UNROLL
                for (int v = 0; v < nv; ++v) sc[v] = var_to_coord(M, v, sq[v]);
                // K5: the table lookup only needs the coordinates; issued here, its probe lands behind the planning-link FK
                if (out_id) { early_id = table_lookup<false>(Sq->table, sc, nv); have_early = true; }
                double p[3];
                planning_fk(M, sq, p);
                bool is_goal;
                if (Sq->goal.type == SMPLX_GOAL_JOINT) {
                    is_goal = true;
                    MV_UNROLL
                    for (int v = 0; v < nv; ++v)
                        if (fabs((double)(sc[v] - Sq->goal.coord[v])) > Sq->goal.angle_tol[v]) is_goal = false;
                } else {
                    is_goal = fabs(p[0] - Sq->goal.xyz[0]) <= Sq->goal.xyz_tol[0] &&
                              fabs(p[1] - Sq->goal.xyz[1]) <= Sq->goal.xyz_tol[1] &&
                              fabs(p[2] - Sq->goal.xyz[2]) <= Sq->goal.xyz_tol[2];
                }
                int c[3];
                world_to_cell(grid, p, c);
                h = bfs_cost_to_goal(bfs, c);
                cost = A.cost[pi];
                flags = SMPLX_F_VALID | (is_goal ? SMPLX_F_GOAL : 0);
            }
        }
        if (!deferred) {
            out_flags[tid] = (unsigned char)flags;
            out_h[tid] = h;
            out_cost[tid] = cost;
            out_lookups[tid] = lookups;
        } else {
            h = out_h[tid];
        }
        succ_h = h;
        // K5: getHashEntry on the device copy of the state table (manip_lattice.cpp:1302-1316).  The id is only a
        // hint to the host (it skips its own lookup); ids are still ASSIGNED on the host, in commit order.
        if (out_id) {
            if (flags & SMPLX_F_VALID) succ_id = have_early ? early_id : table_lookup<false>(Sq->table, out_coord + tid * nv, nv);
            out_id[tid] = succ_id;
        }
    }
    // K5: validity compaction with wavefront ballots.  A valid successor leaves 8 bytes in region A; one whose
    // coordinate the table does not know (or a goal successor, whose own joint values extractPath reports) also a full
    // record in region B.  Ballot -> popcount of the lanes below -> wave totals in LDS -> ONE atomic per region and block.
    if (cmp.rec_a) {
        __shared__ int c_cnt[BLOCK / 64][2];
        __shared__ int c_base[2];
        const bool is_a = in_range && (flags & SMPLX_F_VALID) != 0;
        const bool is_b = is_a && (succ_id < 0 || (flags & SMPLX_F_GOAL) != 0);
        const unsigned long long m_a = __ballot(is_a), m_b = __ballot(is_b);
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
        if (lane == 0) { c_cnt[wv][0] = __popcll(m_a); c_cnt[wv][1] = __popcll(m_b); }
        __syncthreads();
        if (threadIdx.x == 0) {
            int ta = 0, tb = 0;
#pragma unroll
            for (int k = 0; k < BLOCK / 64; ++k) { ta += c_cnt[k][0]; tb += c_cnt[k][1]; }
            const int shard = blockIdx.x % SMPLX_CMP_SHARDS;
            const int sa = cmp.cap_a / SMPLX_CMP_SHARDS, sb = cmp.cap_b / SMPLX_CMP_SHARDS;
            int ba = ta > 0 ? atomicAdd(&cmp.totals[32 * shard], ta) : 0;
            int bb = tb > 0 ? atomicAdd(&cmp.totals[32 * shard + 1], tb) : 0;
            if (ba + ta > sa || bb + tb > sb) { cmp.totals[32 * SMPLX_CMP_SHARDS] = 1; ba = -1; }   // overflow: dense outputs stay valid
            else { ba += shard * sa; bb += shard * sb; }
            c_base[0] = ba; c_base[1] = bb;
            int* bt = cmp.block_tab + 4 * (size_t)blockIdx.x;
            bt[0] = ba; bt[1] = ta; bt[2] = bb; bt[3] = tb;
        }
        __syncthreads();
        if (c_base[0] >= 0 && is_a) {
            int ia = c_base[0] + __popcll(m_a & below), ib = c_base[1] + __popcll(m_b & below);
            for (int k = 0; k < wv; ++k) { ia += c_cnt[k][0]; ib += c_cnt[k][1]; }
            const int si = (int)(tid / nprims);
            const int pi = (int)(tid - (long long)si * nprims);
            cmp.rec_a[2 * (size_t)ia] = succ_id;
            cmp.rec_a[2 * (size_t)ia + 1] = pi | ((flags & SMPLX_F_GOAL) ? 0x100 : 0) | (si << 9);
            if (is_b) {
                const int nv = MV_NVARS(M);
                unsigned char* rb = cmp.rec_b + (size_t)ib * cmp.rec_b_bytes;
                int* ri = (int*)rb;
                double* rq = (double*)(rb + (size_t)((nv + 2) / 2 * 2) * 4);
                ri[0] = succ_h;
                MV_UNROLL
                for (int v = 0; v < nv; ++v) { ri[1 + v] = out_coord[tid * nv + v]; rq[v] = out_q[tid * nv + v]; }
            }
        }
    }
    if (counters) {
        const unsigned long long m_eval = __ballot(evaluated);
        const unsigned long long m_valid = __ballot((flags & SMPLX_F_VALID) != 0);
        tally_block(counters, __popcll(m_eval), __popcll(m_valid), lookups, performed, ncfg, slk);
    }
}

// ---------------------------------------------------------------------------------------------
// Small frontier batches (a search that misses on a handful of states) and the device-resident search (k_search): ONE
// block evaluates ONE state, because at this size the cost is launch + dependency latency, not throughput.
// Every WAVE of the block has one role, so that no wave runs two long code paths one after the other:
//   config waves   lanes 0 .. 7 M - 1: lane (p, k) checks waypoints k+1, k+8, ... of edge p (an edge with more than
//                  7 waypoints after the start wraps around its lanes); lane 7 M: the state itself (waypoint 0 of
//                  every edge).  One configuration per lane, one code path per wave.
//   last wave      lane p < M: the successor of primitive p -- joint values, limits, coordinates, planning-link FK,
//                  goal test, heuristic, and at the end the verdict; lane M: the state's metric goal distance (the gate
//                  of the primitives).
// The goal distance is computed first (one lane, while the successor joint values are formed): only the primitives it
// activates have their waypoints checked -- an ungated snap-to-goal primitive is an edge of a hundred waypoints, 15
// configurations in sequence on each of its 7 lanes (measured in round 2: 104 us per launch instead of 22).
// Results are identical to the pipeline.
// ---------------------------------------------------------------------------------------------
#define SMPLX_SMALL_LANES 7   // waypoint lanes per edge

// what one block-level expansion leaves in LDS (static shared memory of the calling kernel)
struct ExpandLds {
    double goal_dist;
    int state_bad, state_lookups;
    double parent[SMPLX_MAX_VARS];
    double sq[SMPLX_MAX_PRIMS][SMPLX_MAX_VARS];   // successor joint values of every primitive
    int coord[SMPLX_MAX_PRIMS][SMPLX_MAX_VARS];   // defined where flags has the valid bit
    int edge_bad[SMPLX_MAX_PRIMS], edge_lk[SMPLX_MAX_PRIMS];
    int h[SMPLX_MAX_PRIMS], lookups[SMPLX_MAX_PRIMS];
    int flags[SMPLX_MAX_PRIMS];
};

// ---- the GetSuccs loop body (manip_lattice.cpp:254-305) in lane-sized pieces; expand_state_block (k_small_batch) and
// k_search (search_kernel.h) put them together around their own barriers ----

// can the primitive produce an action at all (a snap needs a joint-space goal: manip_lattice_action_space.cpp:551-559)
__device__ __forceinline__ bool prim_has_action(const SmplxActionsDev& A, const SmplxGoalDev& G, int p)
{
    const int ty = A.type[p];
    return ty == SMPLX_MP_LONG || ty == SMPLX_MP_SHORT || (ty == SMPLX_MP_SNAP_XYZ_RPY && G.type == SMPLX_GOAL_JOINT);
}

// bookkeeping lane of primitive p, first half: the successor's joint values -> X.sq[p] (applyMotionPrimitive,
// manip_lattice_action_space.cpp:575-621)
__device__ __forceinline__ void expand_successor_values(const ModelLds* __restrict__ M, const SmplxActionsDev& A,
                                                        const SmplxGoalDev& G, ExpandLds& X, int p)
{
    const int nv = MV_NVARS(M);
    const double* parent = X.parent;
    const int type = A.type[p];
    if (type == SMPLX_MP_LONG || type == SMPLX_MP_SHORT) {
        double d0 = A.delta[p][0], d1 = nv > 1 ? A.delta[p][1] : 0.0;
        if (A.xy_rotate_by_var3 && nv > 3) {
            double sn, cs;
            smplx_sincos(parent[3], &sn, &cs);
            const double a0 = d0, a1 = d1;
            d0 = cs * a0 + (-sn) * a1;
            d1 = sn * a0 + cs * a1;
        }
        MV_UNROLL
        for (int v = 0; v < nv; ++v) {
            const double d = v == 0 ? d0 : (v == 1 ? d1 : A.delta[p][v]);
            X.sq[p][v] = d + parent[v];
        }
    } else {
        MV_UNROLL
        for (int v = 0; v < nv; ++v) X.sq[p][v] = G.angles[v];   // :551-559
    }
}

// metric goal distance of the state in X.parent (bfs_heuristic.cpp:129-138): the gate of its primitives
__device__ __forceinline__ double expand_goal_distance(const ModelLds* __restrict__ M, const SmplxGridDev& grid, const SmplxBfsDev& bfs,
                                                       const ExpandLds& X)
{
    double pw[3];
    planning_fk(M, X.parent, pw);
    int c[3];
    world_to_cell(grid, pw, c);
    return !bfs_in_bounds(bfs, c) ? (double)0x7FFFFFFF * grid.res : (double)bfs_dist(bfs, c) * grid.res;
}

// waypoint count of the edge parent -> sq (robot_motion_collision_model.cpp:371-407, .h:352-366, 173-181)
__device__ __forceinline__ int expand_waypoint_count(const ModelLds* __restrict__ M, const double* parent, const double* sq)
{
    const int nv = MV_NVARS(M);
    double motion = 0.0;
    MV_UNROLL
    for (int v = 0; v < nv; ++v) {
        const int vt = MV_TYPE(M, v);
        const double sv = parent[v], fv = sq[v];
        if (vt == SMPLX_JT_CONTINUOUS) motion += MV_K(M, v) * fabs(smplx_shortest_angle_diff(fv, sv));
        else if (vt == SMPLX_JT_REVOLUTE) motion += MV_K(M, v) * fabs(fv - sv);
        else if (vt == SMPLX_JT_PRISMATIC) motion += fabs(fv - sv);
    }
    int W = 0;
    if (motion != 0.0) {
        W = (int)ceil(motion / 0.05) + 1;
        if (W < 2) W = 2;
    }
    return W;
}

// config lane c of the block: c < ncfg - 1: lane (p, slot) checks waypoints slot+1, slot+8, ... of edge p (an edge longer
// than 7 waypoints wraps around its lanes); c == ncfg - 1: the state itself (waypoint 0 of every edge)
template <bool RS = false>
__device__ __forceinline__ void expand_config_lane(const ModelLds* __restrict__ M, const ThreadLds& L, const SmplxActionsDev& A,
                                                   const SmplxGoalDev& G, const SmplxGridDev& grid, ExpandLds& X, int c, int ncfg)
{
    const double* parent = X.parent;
    if (c < ncfg - 1) {
        const int p = c / SMPLX_SMALL_LANES, slot = c % SMPLX_SMALL_LANES;
        if (prim_has_action(A, G, p) && mprim_active(A, X.goal_dist, A.type[p])) {
            const double* sq = X.sq[p];
            if (check_joint_limits(M, sq)) {
                const int Wc = expand_waypoint_count(M, parent, sq);
                int my_bad = 0, my_lk = 0;
                for (int wp = slot + 1; wp < Wc && !my_bad; wp += SMPLX_SMALL_LANES) {
                    EdgeRef e;
                    e.start = parent; e.finish = sq;
                    e.alpha = (double)wp * (1.0 / (double)(Wc - 1));
                    const bool ok = config_valid<RS>(M, L, grid, e, my_lk);
                    my_bad = ok ? 0 : 1;
                }
                if (my_bad) atomicOr(&X.edge_bad[p], 1);
                if (my_lk) atomicAdd(&X.edge_lk[p], my_lk);
            }
        }
    } else if (c == ncfg - 1) {
        EdgeRef e;
        e.start = parent; e.finish = parent; e.alpha = 0.0;
        int lk = 0;
        const bool ok = config_valid<RS>(M, L, grid, e, lk);
        if (!ok) atomicOr(&X.state_bad, 1);
        if (lk) atomicAdd(&X.state_lookups, lk);
    }
}

// bookkeeping lane of primitive p, second half, in two steps: (i) limits, waypoint count, coordinates (-> X.coord[p]);
// (ii) planning-link FK, goal test, heuristic.  (k_search starts the state-table probe of the coordinate between the two.)
struct BookLane { bool limits_ok; int W, h, is_goal; };
__device__ __forceinline__ void expand_book_coords(const ModelLds* __restrict__ M, ExpandLds& X, int p, BookLane& r)
{
    const int nv = MV_NVARS(M);
    r.W = 0; r.h = 0; r.is_goal = 0;
    const double* sq = X.sq[p];
    r.limits_ok = check_joint_limits(M, sq);
    if (r.limits_ok) {
        r.W = expand_waypoint_count(M, X.parent, sq);
        MV_UNROLL
        for (int v = 0; v < nv; ++v) X.coord[p][v] = var_to_coord(M, v, sq[v]);
    }
}
__device__ __forceinline__ void expand_book_goal(const ModelLds* __restrict__ M, const SmplxGridDev& grid, const SmplxBfsDev& bfs,
                                                 const SmplxGoalDev& G, const ExpandLds& X, int p, BookLane& r)
{
    if (!r.limits_ok) return;
    const int nv = MV_NVARS(M);
    double pw[3];
    planning_fk(M, X.sq[p], pw);
    if (G.type == SMPLX_GOAL_JOINT) {      // manip_lattice.cpp:1596-1606
        r.is_goal = 1;
        MV_UNROLL
        for (int v = 0; v < nv; ++v)
            if (fabs((double)(X.coord[p][v] - G.coord[v])) > G.angle_tol[v]) r.is_goal = 0;
    } else {                                      // XYZ goal :1672-1687
        r.is_goal = fabs(pw[0] - G.xyz[0]) <= G.xyz_tol[0] && fabs(pw[1] - G.xyz[1]) <= G.xyz_tol[1] &&
                    fabs(pw[2] - G.xyz[2]) <= G.xyz_tol[2];
    }
    int c[3];
    world_to_cell(grid, pw, c);
    r.h = bfs_cost_to_goal(bfs, c);
}
__device__ __forceinline__ BookLane expand_book_lane(const ModelLds* __restrict__ M, const SmplxGridDev& grid, const SmplxBfsDev& bfs,
                                                     const SmplxGoalDev& G, ExpandLds& X, int p)
{
    BookLane r;
    expand_book_coords(M, X, p, r);
    expand_book_goal(M, grid, bfs, G, X, p, r);
    return r;
}

// the verdict of edge p once the waypoint lanes have reported: SMPLX_F_* flags; lookups = the reference's tally for the edge
__device__ __forceinline__ int expand_verdict(const SmplxActionsDev& A, const ExpandLds& X, int p, bool have_action, const BookLane& b,
                                              int& lookups)
{
    lookups = 0;
    if (!have_action || !mprim_active(A, X.goal_dist, A.type[p])) return SMPLX_F_INACTIVE;
    if (!b.limits_ok) return SMPLX_F_LIMITS;
    lookups = X.edge_lk[p] + (b.W > 0 ? X.state_lookups : 0);
    const bool ok = (b.W == 0) || (X.state_bad == 0 && X.edge_bad[p] == 0);
    if (!ok) return SMPLX_F_COLLISION;
    return SMPLX_F_VALID | (b.is_goal ? SMPLX_F_GOAL : 0);
}

// lanes of ONE wave exchange data through LDS: no block barrier needed, only that neither the compiler nor the memory
// pipeline reorders the accesses (LDS operations of a wave execute in order)
#define SMPLX_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                               __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// The whole loop body for the state whose joint values are at parent_src (HBM or pinned host memory), by all threads of the
// block (blockDim.x = smplx_small_block(nprims)).  The bookkeeping wave loads the parent itself and starts at once; the
// other waves join at the first of three barriers.  Ends with a barrier: on return X.parent, X.flags, X.sq, X.coord, X.h,
// X.lookups, X.goal_dist, X.state_bad and X.state_lookups are final.
__device__ __forceinline__ void expand_state_block(const ModelLds* __restrict__ M, const ThreadLds& L, const SmplxSpaceDev* __restrict__ S,
                                                   const SmplxSpaceDev* __restrict__ Sq, const SmplxGridDev& grid, ExpandLds& X,
                                                   const double* __restrict__ parent_src)
{
    const SmplxActionsDev& A = S->actions;
    const SmplxBfsDev bfs = Sq->bfs;
    const int nprims = A.nprims, nv = MV_NVARS(M);
    const int t = threadIdx.x;
    const int ncfg = nprims * SMPLX_SMALL_LANES + 1;          // config lanes (the last one: the state itself)
    const int book0 = (ncfg + 63) / 64 * 64;                  // first lane of the bookkeeping wave
    if (t < nprims) { X.edge_bad[t] = 0; X.edge_lk[t] = 0; }
    if (t == 0) { X.state_bad = 0; X.state_lookups = 0; }
    const int bp = t - book0;                                 // primitive of a bookkeeping lane
    const bool book = bp >= 0 && bp < nprims;
    if (bp >= 0) {
        if (bp < nv) X.parent[bp] = parent_src[bp];
        SMPLX_WAVE_SYNC();
    }
    const bool have_action = book && prim_has_action(A, Sq->goal, bp);
    if (have_action) expand_successor_values(M, A, Sq->goal, X, bp);
    if (bp == nprims) X.goal_dist = expand_goal_distance(M, grid, bfs, X);
    __syncthreads();   // every lane of every edge can read its successor's joint values and the gate from LDS
    BookLane b;
    b.limits_ok = false; b.W = 0; b.h = 0; b.is_goal = 0;
    if (t < book0) expand_config_lane(M, L, A, Sq->goal, grid, X, t, ncfg);
    else if (have_action) b = expand_book_lane(M, grid, bfs, Sq->goal, X, bp);
    __syncthreads();   // the waypoint verdicts and the state's own check have landed in LDS
    if (book) {
        int lookups;
        const int flags = expand_verdict(A, X, bp, have_action, b, lookups);
        X.flags[bp] = flags;
        X.h[bp] = (flags & SMPLX_F_VALID) ? b.h : 0;
        X.lookups[bp] = lookups;
    }
    __syncthreads();
}

extern "C" __global__ void __launch_bounds__(512)
k_small_batch(const SmplxSpaceDev* __restrict__ S, const double* __restrict__ Q, const int64_t* __restrict__ refs, int B,
              double* __restrict__ goal_dist_out, unsigned char* __restrict__ state_bad_out, int* __restrict__ state_lookups_out,
              unsigned char* __restrict__ out_flags, int* __restrict__ out_coord, double* __restrict__ out_q,
              int* __restrict__ out_h, int* __restrict__ out_cost, int* __restrict__ out_lookups,
              const SmplxSpaceDev* const* __restrict__ stab, const unsigned short* __restrict__ state_q,
              unsigned char* __restrict__ host_flags, int* __restrict__ host_coord, double* __restrict__ host_q,
              int* __restrict__ host_h, int* __restrict__ out_id, int* __restrict__ host_id,
              const int* __restrict__ ins_items, int n_ins)
{
    // host_*: optional pinned host buffers the results are ALSO written to (zero-copy: a small batch costs less
    // as a few KB of PCIe stores than as DMA copies); Q may itself be pinned host memory -- the parent's
    // joint values are staged into LDS once per block
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ ExpandLds X;
    if (n_ins > 0 && table_insert_block(S, stab, ins_items, n_ins, B)) return;   // K5: see k_pipe_prep
    ModelLds Mv;
    ThreadLds L = setup_lds(S, smem, &Mv, blockDim.x);
    const ModelLds* M = &Mv;
    const SmplxActionsDev& A = S->actions;
    const SmplxGridDev grid = S->grid;
    const long long si = blockIdx.x;
    const SmplxSpaceDev* Sq = stab ? stab[state_q[si]] : S;
    const int nprims = A.nprims, nv = MV_NVARS(M);
    const int t = threadIdx.x;
    const long long parent_at = refs ? refs[si] : (int64_t)si;      // where the parent's joint values sit in Q (units of nv)
    expand_state_block(M, L, S, Sq, grid, X, Q + parent_at * nv);
    if (t == 0) { goal_dist_out[si] = X.goal_dist; state_bad_out[si] = (unsigned char)X.state_bad; state_lookups_out[si] = X.state_lookups; }
    if (t < nprims) {
        const long long eid = si * nprims + t;
        const int flags = X.flags[t];
        out_flags[eid] = (unsigned char)flags;
        out_h[eid] = X.h[t];
        out_cost[eid] = (flags & SMPLX_F_VALID) ? A.cost[t] : 0;
        out_lookups[eid] = X.lookups[t];
        const bool active = !(flags & SMPLX_F_INACTIVE);
        if (active) {
            MV_UNROLL
            for (int v = 0; v < nv; ++v) out_q[eid * nv + v] = X.sq[t][v];
        }
        int sid = -1;
        if (flags & SMPLX_F_VALID) {
            MV_UNROLL
            for (int v = 0; v < nv; ++v) out_coord[eid * nv + v] = X.coord[t][v];
            if (out_id) sid = table_lookup<true>(Sq->table, X.coord[t], nv);   // K5: device copy of the state table (see k_pipe_finish)
        }
        if (out_id) out_id[eid] = sid;
        if (host_flags) {
            host_flags[eid] = (unsigned char)flags;
            if (host_id) host_id[eid] = sid;
            if (flags & SMPLX_F_VALID) {
                host_h[eid] = X.h[t];
                MV_UNROLL
                for (int v = 0; v < nv; ++v) { host_coord[eid * nv + v] = X.coord[t][v]; host_q[eid * nv + v] = X.sq[t][v]; }
            }
        }
    }
}

extern "C" __global__ void __launch_bounds__(BLOCK, 2)   // >= 2 waves per SIMD: at most 256 VGPRs, whichever compiler builds it
k_edge_valid(const SmplxSpaceDev* __restrict__ S, const double* __restrict__ Aq, const double* __restrict__ Bq, int n,
             unsigned char* __restrict__ out, int* __restrict__ out_lookups, int* __restrict__ out_waypoints)
{
    extern __shared__ __align__(16) unsigned char smem[];
    ModelLds Mv;
#ifdef SMPLX_CONST_MODEL
    constexpr bool RS = true;      // as k_state_valid below
#else
    constexpr bool RS = false;
#endif
    ThreadLds L = setup_lds(S, smem, &Mv, BLOCK, !RS);
    const ModelLds* M = &Mv;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    int lk = 0, W = 0;
    const SmplxGridDev grid = S->grid;
    const bool ok = edge_valid<RS>(M, L, grid, Aq + (size_t)i * MV_NVARS(M), Bq + (size_t)i * MV_NVARS(M), false, true, lk, W);
    out[i] = ok ? 1 : 0;
    if (out_lookups) out_lookups[i] = lk;
    if (out_waypoints) out_waypoints[i] = W;
}

extern "C" __global__ void __launch_bounds__(BLOCK, 2)   // >= 2 waves per SIMD: at most 256 VGPRs, whichever compiler builds it
k_state_valid(const SmplxSpaceDev* __restrict__ S, const double* __restrict__ Q, int n, unsigned char* __restrict__ out,
              int* __restrict__ out_lookups)
{
    extern __shared__ __align__(16) unsigned char smem[];
    ModelLds Mv;
#ifdef SMPLX_CONST_MODEL
    constexpr bool RS = true;      // saved link transforms in registers: engine.hip sizes this kernel's LDS without the slots
#else
    constexpr bool RS = false;
#endif
    ThreadLds L = setup_lds(S, smem, &Mv, BLOCK, !RS);
    const ModelLds* M = &Mv;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    EdgeRef e;
    e.start = Q + (size_t)i * MV_NVARS(M); e.finish = e.start; e.alpha = 0.0;
    int lk = 0;
    const SmplxGridDev grid = S->grid;
    const bool ok = config_valid<RS>(M, L, grid, e, lk);
    out[i] = ok ? 1 : 0;
    if (out_lookups) out_lookups[i] = lk;
}

extern "C" __global__ void __launch_bounds__(BLOCK)
k_heuristic(const SmplxSpaceDev* __restrict__ S, const double* __restrict__ Q, int n, int* __restrict__ out_h,
            double* __restrict__ out_xyz)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const ModelLds Mv = setup_model_only(S, smem);
    const ModelLds* M = &Mv;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    double p[3];
    planning_fk(M, Q + (size_t)i * MV_NVARS(M), p);
    int c[3];
    world_to_cell(S->grid, p, c);
    out_h[i] = bfs_cost_to_goal(S->bfs, c);
    if (out_xyz) { out_xyz[3 * i] = p[0]; out_xyz[3 * i + 1] = p[1]; out_xyz[3 * i + 2] = p[2]; }
}

// BfsHeuristic::getMetricGoalDistance (bfs_heuristic.cpp:129-138) for a batch of workspace points
extern "C" __global__ void __launch_bounds__(BLOCK)
k_bfs_metric(SmplxGridDev grid, SmplxBfsDev bfs, const double* __restrict__ xyz, int n, double* __restrict__ out)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const double p[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    int c[3];
    world_to_cell(grid, p, c);
    out[i] = !bfs_in_bounds(bfs, c) ? (double)0x7FFFFFFF * grid.res : (double)bfs_dist(bfs, c) * grid.res;
}

// debug/parity: world positions of every tree node for one configuration per thread
extern "C" __global__ void __launch_bounds__(BLOCK)
k_sphere_positions(const SmplxSpaceDev* __restrict__ S, const double* __restrict__ Q, int n, double* __restrict__ out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    ModelLds Mv;
    ThreadLds L = setup_lds(S, smem, &Mv);
    const ModelLds* M = &Mv;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const double* q = Q + (size_t)i * MV_NVARS(M);
    double T[12];
    for (int k = 0; k < 12; ++k) T[k] = 0.0;
    for (int j = 0; j < M->njoints; ++j) {
        JointPtr jt = &M->joints[j];
        if (jt->src >= 0) for (int k = 0; k < 12; ++k) T[k] = lds_d(L, L.slot_base + 12 * jt->src + k);
        apply_joint(jt, jt->var >= 0 ? q[jt->var] : 0.0, T, jt->src == SMPLX_SRC_ROOT);
        if (jt->save_slot >= 0) for (int k = 0; k < 12; ++k) lds_d(L, L.slot_base + 12 * jt->save_slot + k) = T[k];
        if (jt->tree >= 0) {
            for (int nd = M->tree_first[jt->tree]; nd < M->tree_first[jt->tree + 1]; ++nd) {
                double c[3] = {L.nodes[nd].c[0], L.nodes[nd].c[1], L.nodes[nd].c[2]};
                double p[3];
                xform(T, c, p);
                double* o = out + ((size_t)i * M->nnodes + nd) * 3;
                o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// BFS-3D (bfs3d.cpp:156-201 run, 507-547 search), brick formulation over brick-major records (device_types.h
// SmplxBfsDev).  A wave owns an 8x8x8 brick: it loads the brick and its one-cell halo into LDS ONCE, relaxes
// d(c) = min(d(c), min over the 26 neighbours of d + 1) until nothing changes, writes the record back if anything improved,
// and queues the neighbour bricks whose halo it changed.  A pass runs over the queued bricks; passes repeat until none is
// queued.  Hop counts with unit edge costs are the unique fixed point of that relaxation from d(goal) = 0, so the grid equals
// the sequential queue's (bfs3d.cpp:507-547) whatever order bricks are visited in: a brick that read a neighbour's old value
// is queued again by that neighbour when the value drops.  (Round 1 ran one launch per BFS level, 192 at 256^3; round 2 the
// bricks over the reference's x-fastest array.)
// Sentinels as in the reference: WALL 0x7FFFFFFF never changes, UNDISCOVERED -1 stays -1 where no path leads.
// ---------------------------------------------------------------------------------------------
#define SMPLX_BFS_INF 0x7FFFFFFEu
#define SMPLX_BFS_WALLV 0xFFFFFFFEu     // (not 0xFFFFFFFF: a wall + 1 must not wrap to 0 in the branch-free relaxation)
#define SMPLX_BFS_SHARDS 16      // the brick list of a pass is cut into sub-lists with their counters on separate 128-byte lines

#define SMPLX_BRICK 8
#define SMPLX_BRICK_TILE (SMPLX_BRICK + 2)

// local cell of slot s < SMPLX_BFS_USED of a record (interior, face copy, edge copy, or -- with a coordinate of -1 or 8 -- the
// diagonal neighbour's cell next to a corner)
__device__ __forceinline__ void bfs_slot_cell(int s, int& lx, int& ly, int& lz)
{
    if (s < SMPLX_BFS_FACES) { lz = s >> 6; ly = (s >> 3) & 7; lx = s & 7; return; }
    if (s < SMPLX_BFS_EDGES) {
        const int f = (s - SMPLX_BFS_FACES) >> 6, a = ((s - SMPLX_BFS_FACES) >> 3) & 7, c = (s - SMPLX_BFS_FACES) & 7;
        if (f < 2) { lx = f == 0 ? 0 : 7; lz = a; ly = c; }
        else if (f < 4) { ly = f == 2 ? 0 : 7; lz = a; lx = c; }
        else { lz = f == 4 ? 0 : 7; ly = a; lx = c; }
        return;
    }
    if (s < SMPLX_BFS_CORNERS) {
        const int k = (s - SMPLX_BFS_EDGES) >> 3;
        lz = (s - SMPLX_BFS_EDGES) & 7;
        lx = (k & 1) ? 7 : 0;
        ly = (k & 2) ? 7 : 0;
        return;
    }
    const int c = s - SMPLX_BFS_CORNERS;      // a neighbour's cell, one step outside the brick
    lx = (c & 1) ? 8 : -1;
    ly = (c & 2) ? 8 : -1;
    lz = (c & 4) ? 8 : -1;
}

// every slot of a record that holds cell (lx, ly, lz) takes v
__device__ __forceinline__ void bfs_record_store_cell(int* __restrict__ rec, int lx, int ly, int lz, int v)
{
    rec[(lz << 6) + (ly << 3) + lx] = v;
    if (lx == 0) rec[SMPLX_BFS_FACES + 0 * 64 + lz * 8 + ly] = v;
    if (lx == 7) rec[SMPLX_BFS_FACES + 1 * 64 + lz * 8 + ly] = v;
    if (ly == 0) rec[SMPLX_BFS_FACES + 2 * 64 + lz * 8 + lx] = v;
    if (ly == 7) rec[SMPLX_BFS_FACES + 3 * 64 + lz * 8 + lx] = v;
    if (lz == 0) rec[SMPLX_BFS_FACES + 4 * 64 + ly * 8 + lx] = v;
    if (lz == 7) rec[SMPLX_BFS_FACES + 5 * 64 + ly * 8 + lx] = v;
    if ((lx == 0 || lx == 7) && (ly == 0 || ly == 7)) rec[SMPLX_BFS_EDGES + (((ly == 7) ? 2 : 0) + ((lx == 7) ? 1 : 0)) * 8 + lz] = v;
}

// a CORNER cell of brick (bx, by, bz) also goes into the corner slot of the brick diagonally across it (device_types.h)
__device__ __forceinline__ void bfs_push_corner(int* __restrict__ dist, int bx, int by, int bz, int nbx, int nby, int nbz,
                                                int lx, int ly, int lz, int v)
{
    const int qx = bx + (lx == 7 ? 1 : -1), qy = by + (ly == 7 ? 1 : -1), qz = bz + (lz == 7 ? 1 : -1);
    if (qx < 0 || qy < 0 || qz < 0 || qx >= nbx || qy >= nby || qz >= nbz) return;
    // seen from there this brick lies on the low side of an axis where the cell is at 7
    const int c = (lx == 7 ? 0 : 1) | ((ly == 7 ? 0 : 1) << 1) | ((lz == 7 ? 0 : 1) << 2);
    dist[(size_t)((qz * nby + qy) * nbx + qx) * SMPLX_BFS_REC + SMPLX_BFS_CORNERS + c] = v;
}

// The halo of a brick beyond its six faces: 12 edges of 8 cells and 8 corners, piece e < 104.  Where piece e comes from (the
// neighbour brick (ddx, ddy, ddz), the slot of that brick's record) and where it sits in the 10x10x10 tile: a z-parallel
// edge from the neighbour's edge copies, an x- or y-parallel one from the face copy whose fastest index runs along it, a
// corner from the corner slots of the brick's own record (the diagonal neighbours keep them current: bfs_push_corner).
// Functions of e alone: a lane works them out once per launch.
__device__ __forceinline__ void bfs_edge_piece(int e, int& ddx, int& ddy, int& ddz, int& src, int& pos)
{
    constexpr int TL = SMPLX_BRICK_TILE, TP = SMPLX_BRICK_TILE * SMPLX_BRICK_TILE, HI = SMPLX_BRICK_TILE - 1;
    const int i = e & 7, k = (e >> 3) & 3, g = e >> 5;
    const int s0 = k & 1, s1 = k >> 1;          // 0: the low side (neighbour at -1, its cell 7), 1: the high side
    if (g == 0) {            // z runs; (x, y) sides s0, s1
        ddx = s0 ? 1 : -1; ddy = s1 ? 1 : -1; ddz = 0;
        src = SMPLX_BFS_EDGES + ((s1 ? 0 : 2) + (s0 ? 0 : 1)) * 8 + i;
        pos = (i + 1) * TP + (s1 ? HI : 0) * TL + (s0 ? HI : 0);
    } else if (g == 1) {     // x runs; (y, z) sides s0, s1: the neighbour's y face copy
        ddx = 0; ddy = s0 ? 1 : -1; ddz = s1 ? 1 : -1;
        src = SMPLX_BFS_FACES + (s0 ? 2 : 3) * 64 + (s1 ? 0 : 7) * 8 + i;
        pos = (s1 ? HI : 0) * TP + (s0 ? HI : 0) * TL + (i + 1);
    } else if (g == 2) {     // y runs; (x, z) sides s0, s1: the neighbour's x face copy
        ddx = s0 ? 1 : -1; ddy = 0; ddz = s1 ? 1 : -1;
        src = SMPLX_BFS_FACES + (s0 ? 0 : 1) * 64 + (s1 ? 0 : 7) * 8 + i;
        pos = (s1 ? HI : 0) * TP + (i + 1) * TL + (s0 ? HI : 0);
    } else {                 // corners: e = 96 + (cx | cy << 1 | cz << 2), from the corner slots of the brick's OWN record
        const int cx = i & 1, cy = (i >> 1) & 1, cz = (i >> 2) & 1;
        ddx = 0; ddy = 0; ddz = 0;
        src = SMPLX_BFS_CORNERS + i;
        pos = (cz ? HI : 0) * TP + (cy ? HI : 0) * TL + (cx ? HI : 0);
    }
}

// walls: BfsHeuristic::syncGridAndBfs (bfs_heuristic.cpp:331-353) in integer form:
// wall iff squared cell distance <= wall_thr (largest i with res*sqrt(i) <= radius; -1 if none)
extern "C" __global__ void __launch_bounds__(256)
k_bfs_init(SmplxGridDev g, int wall_thr, int nbx, int nby, int nbz, int* __restrict__ dist)
{
    const size_t total = (size_t)nbx * nby * nbz * SMPLX_BFS_REC;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int s = (int)(i % SMPLX_BFS_REC);
        const size_t b = i / SMPLX_BFS_REC;
        int v = 0x7FFFFFFF;
        if (s < SMPLX_BFS_USED) {
            int lx, ly, lz;
            bfs_slot_cell(s, lx, ly, lz);
            const int cx = (int)(b % nbx) * 8 + lx, cy = (int)(b / nbx % nby) * 8 + ly, cz = (int)(b / ((size_t)nbx * nby)) * 8 + lz;
            if (cx >= 0 && cy >= 0 && cz >= 0 && cx < g.n[0] && cy < g.n[1] && cz < g.n[2]) {      // (a corner slot can lie outside the grid: a wall)
                const size_t brick = ((size_t)(cx >> 2) * g.bricks[1] + (cy >> 2)) * g.bricks[2] + (cz >> 2);
                const int d2 = (int)g.d2[brick * 64 + ((cx & 3) << 4) + ((cy & 3) << 2) + (cz & 3)];
                v = d2 <= wall_thr ? 0x7FFFFFFF : -1;
            }
        }
        dist[i] = v;
    }
}

// BFS_3D::run reset: every non-wall cell back to UNDISCOVERED (bfs3d.cpp:162-166)
extern "C" __global__ void __launch_bounds__(256)
k_bfs_reset(int* __restrict__ dist, size_t total)
{
    for (size_t node = (size_t)blockIdx.x * 256 + threadIdx.x; node < total; node += (size_t)gridDim.x * 256)
        if (dist[node] != 0x7FFFFFFF) dist[node] = -1;
}

// the padded (nx+2)(ny+2)(nz+2) grid in the reference's node order (bfs3d.h:213-220), for smplx_bfs_copy
extern "C" __global__ void __launch_bounds__(256)
k_bfs_export(SmplxBfsDev b, int* __restrict__ out)
{
    const size_t total = (size_t)b.dim_x * b.dim_y * b.dim_z;
    for (size_t node = (size_t)blockIdx.x * 256 + threadIdx.x; node < total; node += (size_t)gridDim.x * 256) {
        const int x = (int)(node % b.dim_x), y = (int)(node / b.dim_x % b.dim_y), z = (int)(node / ((size_t)b.dim_x * b.dim_y));
        int v = 0x7FFFFFFF;
        if (!(x == 0 || x == b.dim_x - 1 || y == 0 || y == b.dim_y - 1 || z == 0 || z == b.dim_z - 1)) {
            const int c[3] = {x - 1, y - 1, z - 1};
            v = bfs_dist(b, c);
        }
        out[node] = v;
    }
}

extern "C" __global__ void __launch_bounds__(64)
k_bfs_brick_seed(int* __restrict__ dist, int cx, int cy, int cz, int nbx, int nby, int nbz, int* __restrict__ list0, int* __restrict__ counts, int tag_word)
{
    // counts: 3 sets x SMPLX_BFS_SHARDS counters, 32 ints apart
    if (blockIdx.x == 0) {
        const int t = threadIdx.x;
        if (t < 3 * SMPLX_BFS_SHARDS) counts[32 * t] = 0;
        __syncthreads();
        if (t == 0) {
            const int brick = ((cz >> 3) * nby + (cy >> 3)) * nbx + (cx >> 3);
            bfs_record_store_cell(dist + (size_t)brick * SMPLX_BFS_REC, cx & 7, cy & 7, cz & 7, tag_word);   // distance 0; overwrites a wall at the goal cell, as bfs3d.cpp:178 does
            const int lx = cx & 7, ly = cy & 7, lz = cz & 7;
            if ((lx == 0 || lx == 7) && (ly == 0 || ly == 7) && (lz == 0 || lz == 7)) bfs_push_corner(dist, cx >> 3, cy >> 3, cz >> 3, nbx, nby, nbz, lx, ly, lz, tag_word);
            list0[0] = brick;      // sub-list 0 of list 0
            counts[0] = 1;
        }
    }
}

// One WAVE per brick (block = 64 lanes): lane (x, y) keeps its z-column of 8 cells in registers: no block barriers, many bricks
// resident per CU, and a change travels the whole column within one sweep (the z direction is relaxed in place, up and
// down), so a brick needs a third of the sweeps of a cell-per-thread version.  In-plane neighbours come from the LDS tile, which
// the lanes refresh with their columns at the start of every sweep; the halo (neighbour bricks' cells) is read once and
// never changes during the sweeps.
// The pass builds the next pass's brick list itself: a neighbour brick is claimed with an atomic exchange on its "queued for the
// next pass" word and appended by the claimer.  Two queued-arrays alternate: a brick clears its own word of the array it was
// queued in, so that array is clean again when it next serves as "next".  Three counter sets rotate (in / next / the one zeroed
// for the pass after).
// Lane shifts of the brick kernel: lane t = ty * 8 + tx holds the z-column at (tx, ty).  A lane outside the brick's 8x8
// contributes SMPLX_BFS_WALLV, which never wins a minimum.
struct BfsLanes { bool has_left, has_right, has_up, has_down; int up_addr, down_addr; };
__device__ __forceinline__ unsigned int bfs_min(unsigned int a, unsigned int b) { return a < b ? a : b; }
// min over the lane and its x neighbours (DPP row_shr:1 / row_shl:1 within the 16-lane row: rows of 8 never straddle one)
__device__ __forceinline__ unsigned int bfs_window_x(const BfsLanes& W, unsigned int x)
{
    const unsigned int l = (unsigned int)__builtin_amdgcn_update_dpp((int)SMPLX_BFS_WALLV, (int)x, 0x111, 0xf, 0xf, false);
    const unsigned int r = (unsigned int)__builtin_amdgcn_update_dpp((int)SMPLX_BFS_WALLV, (int)x, 0x101, 0xf, 0xf, false);
    return bfs_min(x, bfs_min(W.has_left ? l : SMPLX_BFS_WALLV, W.has_right ? r : SMPLX_BFS_WALLV));
}
// min over the lane and its y neighbours (lanes t - 8 and t + 8)
__device__ __forceinline__ unsigned int bfs_window_y(const BfsLanes& W, unsigned int x)
{
    const unsigned int u = (unsigned int)__builtin_amdgcn_ds_bpermute(W.up_addr, (int)x);
    const unsigned int d = (unsigned int)__builtin_amdgcn_ds_bpermute(W.down_addr, (int)x);
    return bfs_min(x, bfs_min(W.has_up ? u : SMPLX_BFS_WALLV, W.has_down ? d : SMPLX_BFS_WALLV));
}

#ifdef SMPLX_BFS_TRACE
// diagnostic build (tools/bfs_trace.sh): every visit leaves the 100 MHz wall clock at its phase boundaries; the longest and
// the sum of each phase over the visits of a pass go to g_bfs_trace[k] / [8 + k], the number of visits to [7]
__device__ long long g_bfs_trace[16];
#define BFS_MARK(k) do { if (t == 0) { const long long now_ = (long long)wall_clock64(); \
    if ((k) > 0) { atomicMax((unsigned long long*)&g_bfs_trace[k], (unsigned long long)(now_ - mark_)); atomicAdd((unsigned long long*)&g_bfs_trace[8 + (k)], (unsigned long long)(now_ - mark_)); } \
    else atomicAdd((unsigned long long*)&g_bfs_trace[7], 1ull); \
    mark_ = now_; } } while (0)
#else
#define BFS_MARK(k) do { } while (0)
#endif

extern "C" __global__ void __launch_bounds__(64)
k_bfs_brick_wave(int* __restrict__ dist, int nbx, int nby, int nbz,
                 const int* __restrict__ list_in, const int* __restrict__ counts_in, int* __restrict__ list_next,
                 int* __restrict__ counts_next, int* __restrict__ counts_after, int shard_cap,
                 int* __restrict__ queued_mine, int* __restrict__ queued_next, int* __restrict__ queue_size_out, int tag_word, int tag_mask)
{
    constexpr int TL = SMPLX_BRICK_TILE, TP = SMPLX_BRICK_TILE * SMPLX_BRICK_TILE;
    __shared__ unsigned int tile[TL * TL * TL];
    int pre[SMPLX_BFS_SHARDS + 1];
    pre[0] = 0;
#pragma unroll
    for (int k = 0; k < SMPLX_BFS_SHARDS; ++k) pre[k + 1] = pre[k] + counts_in[32 * k];
    const int n = pre[SMPLX_BFS_SHARDS];
    const int t = threadIdx.x;
    if (blockIdx.x == 0 && t < SMPLX_BFS_SHARDS) counts_after[32 * t] = 0;
    if (blockIdx.x == 0 && t == 0 && queue_size_out) *queue_size_out = n;   // the host sizes the next goal's launches by it
    const int tx = t & 7, ty = t >> 3;
    // the two edge / corner pieces of this lane: pieces t and 64 + t (104 in all)
    int piece_from[2], piece_pos[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        int ddx, ddy, ddz, src, pos;
        bfs_edge_piece(t + 64 * k < 104 ? t + 64 * k : 0, ddx, ddy, ddz, src, pos);
        piece_from[k] = (ddx + 1) | ((ddy + 1) << 2) | ((ddz + 1) << 4) | (src << 6);
        piece_pos[k] = t + 64 * k < 104 ? pos : -1;
    }
#ifdef SMPLX_BFS_TRACE
    long long mark_ = 0;
#endif
    for (int it = blockIdx.x; it < n; it += gridDim.x) {
        int sh = 0;
#pragma unroll
        for (int k = 1; k < SMPLX_BFS_SHARDS; ++k) sh += it >= pre[k] ? 1 : 0;
        BFS_MARK(0);
        const int b = list_in[(size_t)sh * shard_cap + (it - pre[sh])];
        const int bxx = b % nbx, byy = (b / nbx) % nby, bzz = b / (nbx * nby);
        if (t == 0) queued_mine[b] = 0;
        BFS_MARK(1);
        unsigned int v[SMPLX_BRICK], before[SMPLX_BRICK];
        {
            // Sixteen loads a lane, all in flight before the first is used and none inside a branch (a load in a branch is
            // waited for before the branches rejoin).  The lane's own column: eight words of the brick's record, 256 bytes
            // apart -- straight to registers.  The halo goes through the tile: six faces, each one 64-word run of a
            // neighbour's face copy; the edges and corners, 104 words, as pieces (bfs_edge_piece).  A neighbour beyond the grid
            // reads as walls.
            const int own = b * SMPLX_BFS_REC;      // (int: up to 2 M bricks)
            int raw[SMPLX_BRICK], face[6], piece[2];
#pragma unroll
            for (int z = 0; z < SMPLX_BRICK; ++z) raw[z] = dist[own + 64 * z + t];
#pragma unroll
            for (int f = 0; f < 6; ++f) {
                const int d = (f & 1) ? 1 : -1;
                const int qx = bxx + (f < 2 ? d : 0), qy = byy + (f >= 2 && f < 4 ? d : 0), qz = bzz + (f >= 4 ? d : 0);
                const bool ok = !(qx < 0 || qy < 0 || qz < 0 || qx >= nbx || qy >= nby || qz >= nbz);      // (uniform)
                const int w = dist[ok ? ((qz * nby + qy) * nbx + qx) * SMPLX_BFS_REC + SMPLX_BFS_FACES + (f ^ 1) * 64 + t : own];
                face[f] = ok ? w : 0x7FFFFFFF;
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int qx = bxx + ((piece_from[k] & 3) - 1), qy = byy + (((piece_from[k] >> 2) & 3) - 1), qz = bzz + (((piece_from[k] >> 4) & 3) - 1);
                const bool ok = !(qx < 0 || qy < 0 || qz < 0 || qx >= nbx || qy >= nby || qz >= nbz);
                const int w = dist[ok ? ((qz * nby + qy) * nbx + qx) * SMPLX_BFS_REC + (piece_from[k] >> 6) : own];
                piece[k] = ok ? w : 0x7FFFFFFF;
            }
            auto decode = [&](int w) {
                return w == 0x7FFFFFFF ? SMPLX_BFS_WALLV : ((((w ^ tag_word) & tag_mask) != 0 || w == -1) ? SMPLX_BFS_INF : (unsigned int)(w & ~tag_mask));
            };
#pragma unroll
            for (int z = 0; z < SMPLX_BRICK; ++z) { v[z] = decode(raw[z]); before[z] = v[z]; }
            const int a8 = t >> 3, c8 = t & 7;
#pragma unroll
            for (int f = 0; f < 6; ++f) {
                const int side = (f & 1) ? TL - 1 : 0;
                const int pos = f < 2 ? (a8 + 1) * TP + (c8 + 1) * TL + side : (f < 4 ? (a8 + 1) * TP + side * TL + (c8 + 1) : side * TP + (a8 + 1) * TL + (c8 + 1));
                tile[pos] = decode(face[f]);
            }
            tile[piece_pos[0]] = decode(piece[0]);
            if (piece_pos[1] >= 0) tile[piece_pos[1]] = decode(piece[1]);
        }
        __syncthreads();
        BFS_MARK(2);
        const int col = (ty + 1) * TL + (tx + 1);   // this lane's column in a tile plane
        const bool x_lo = tx == 0, x_hi = tx == SMPLX_BRICK - 1, y_lo = ty == 0, y_hi = ty == SMPLX_BRICK - 1;
        const int xh = x_lo ? 0 : TL - 1, yh = y_lo ? 0 : TL - 1;    // the halo column / row beside a boundary lane
        // in-plane 3x3 minima of the two halo planes, and the least halo cell among the in-plane neighbours of every level of
        // a boundary lane's column: the halo does not change during the visit, so these are read once
        unsigned int p_lo = SMPLX_BFS_WALLV, p_hi = SMPLX_BFS_WALLV;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const unsigned int a = tile[col + dy * TL + dx], c = tile[(TL - 1) * TP + col + dy * TL + dx];
                p_lo = a < p_lo ? a : p_lo;
                p_hi = c < p_hi ? c : p_hi;
            }
        unsigned int halo_min[SMPLX_BRICK];
#pragma unroll
        for (int z = 0; z < SMPLX_BRICK; ++z) halo_min[z] = SMPLX_BFS_WALLV;
        if (x_lo || x_hi) {
#pragma unroll
            for (int z = 0; z < SMPLX_BRICK; ++z)
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy) halo_min[z] = bfs_min(halo_min[z], tile[(z + 1) * TP + (ty + 1 + dy) * TL + xh]);
        }
        if (y_lo || y_hi) {
#pragma unroll
            for (int z = 0; z < SMPLX_BRICK; ++z)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) halo_min[z] = bfs_min(halo_min[z], tile[(z + 1) * TP + yh * TL + (tx + 1 + dx)]);
        }
        // The sweeps run in registers: the in-plane neighbours of a column are the columns of lanes t -+ 1 (DPP row shifts) and
        // t -+ 8 (ds_bpermute), 3x3 = a row window then a window of row windows.  (Through the LDS tile -- 64 reads, 8 writes and
        // two barriers a sweep -- a sweep took 1.2 us and the nine or so of a brick the front crosses half its visit.)
        const BfsLanes W = {!x_lo, !x_hi, !y_lo, !y_hi, ((t - 8) & 63) << 2, ((t + 8) & 63) << 2};
        bool is_wall[SMPLX_BRICK];
#pragma unroll
        for (int z = 0; z < SMPLX_BRICK; ++z) is_wall[z] = v[z] == SMPLX_BFS_WALLV;
        while (true) {
            // 3x3 in-plane minimum of every level (own cell included): all row windows, then all the shifts of them in flight
            // together, then the halo's share
            unsigned int pm[SMPLX_BRICK], up[SMPLX_BRICK], down[SMPLX_BRICK], start[SMPLX_BRICK];
#pragma unroll
            for (int z = 0; z < SMPLX_BRICK; ++z) { start[z] = v[z]; pm[z] = bfs_window_x(W, v[z]); }
#pragma unroll
            for (int z = 0; z < SMPLX_BRICK; ++z) {
                up[z] = (unsigned int)__builtin_amdgcn_ds_bpermute(W.up_addr, (int)pm[z]);
                down[z] = (unsigned int)__builtin_amdgcn_ds_bpermute(W.down_addr, (int)pm[z]);
            }
#pragma unroll
            for (int z = 0; z < SMPLX_BRICK; ++z)
                pm[z] = bfs_min(bfs_min(pm[z], halo_min[z]), bfs_min(W.has_up ? up[z] : SMPLX_BFS_WALLV, W.has_down ? down[z] : SMPLX_BFS_WALLV));
            // relax the column in place, upwards then downwards: a cell takes 1 + the least of the three plane minima.  No
            // branches (fifteen divergent ones per sweep were most of its time): an undiscovered or wall minimum + 1 stays
            // above every cell value, a wall keeps its own value by a select
#pragma unroll
            for (int z = 0; z < SMPLX_BRICK; ++z) {
                const unsigned int lo = z == 0 ? p_lo : pm[z - 1], hi = z == SMPLX_BRICK - 1 ? p_hi : pm[z + 1];
                const unsigned int nv = bfs_min(v[z], bfs_min(bfs_min(lo, hi), pm[z]) + 1u);
                v[z] = is_wall[z] ? v[z] : nv;
                pm[z] = bfs_min(pm[z], v[z]);
            }
#pragma unroll
            for (int z = SMPLX_BRICK - 2; z >= 0; --z) {
                const unsigned int nv = bfs_min(v[z], pm[z + 1] + 1u);
                v[z] = is_wall[z] ? v[z] : nv;
                pm[z] = bfs_min(pm[z], v[z]);
            }
            unsigned int diff = 0;
#pragma unroll
            for (int z = 0; z < SMPLX_BRICK; ++z) diff |= start[z] ^ v[z];
            if (__ballot(diff != 0u) == 0ull) break;
        }
        BFS_MARK(3);
        // ---- what improved goes back at once, with its face and edge copies (stores are not waited for) ----
        int* rec = dist + (size_t)b * SMPLX_BFS_REC;
        bool improved = false;
#pragma unroll
        for (int z = 0; z < SMPLX_BRICK; ++z) {
            if (!(v[z] < before[z])) continue;
            improved = true;
            const int val = (int)v[z] | tag_word;
            rec[(z << 6) + (ty << 3) + tx] = val;
            if (x_lo) rec[SMPLX_BFS_FACES + 0 * 64 + z * 8 + ty] = val;
            if (x_hi) rec[SMPLX_BFS_FACES + 1 * 64 + z * 8 + ty] = val;
            if (y_lo) rec[SMPLX_BFS_FACES + 2 * 64 + z * 8 + tx] = val;
            if (y_hi) rec[SMPLX_BFS_FACES + 3 * 64 + z * 8 + tx] = val;
            if (z == 0) rec[SMPLX_BFS_FACES + 4 * 64 + ty * 8 + tx] = val;
            if (z == SMPLX_BRICK - 1) rec[SMPLX_BFS_FACES + 5 * 64 + ty * 8 + tx] = val;
            if ((x_lo || x_hi) && (y_lo || y_hi)) {
                rec[SMPLX_BFS_EDGES + ((y_hi ? 2 : 0) + (x_hi ? 1 : 0)) * 8 + z] = val;
                if (z == 0 || z == SMPLX_BRICK - 1) bfs_push_corner(dist, bxx, byy, bzz, nbx, nby, nbz, tx, ty, z, val);
            }
        }
        // ---- Which neighbour bricks have to look again: only one that CAN improve -- a cell c' of it (this brick's halo holds
        // its value h as of the load; it can only have become smaller since) next to a cell c of this brick with h > d(c) + 1.
        // (Queueing every neighbour that merely SEES a changed cell made the front revisit the bricks behind and beside it: 2.4
        // visits per brick, most of them a load, one sweep and nothing to write.)  A cell c that did not change in this visit
        // cannot pass the test against a current h (the neighbour was queued when c got its value and has read it since), so
        // "changed" need not be tracked.  A halo cell's neighbours in this brick are a window of the brick's boundary layer
        // -- 3x3 for a face, 3 for an edge, 1 for a corner -- so the test is h against the window minimum, the windows built
        // from the columns by the same lane shifts as the sweeps; the halo values come from the tile (they are as loaded).
        // One bit per direction (oz * 9 + oy * 3 + ox, o = 0 / 1 / 2).
        BFS_MARK(4);
        unsigned int mask = 0;
        if (__ballot(improved) != 0ull) {
            auto can_improve = [](unsigned int h, unsigned int w) { return h != SMPLX_BFS_WALLV && w < SMPLX_BFS_INF && h > w + 1u; };
            const int ox = x_lo ? 0 : 2, oy = y_lo ? 0 : 2;
            const bool on_x = x_lo || x_hi, on_y = y_lo || y_hi;
            unsigned int zw[SMPLX_BRICK];      // window along the column
#pragma unroll
            for (int z = 0; z < SMPLX_BRICK; ++z) {
                zw[z] = v[z];
                if (z > 0) zw[z] = bfs_min(zw[z], v[z - 1]);
                if (z < SMPLX_BRICK - 1) zw[z] = bfs_min(zw[z], v[z + 1]);
            }
            bool fx = false, fy = false, exy = false;
#pragma unroll
            for (int z = 0; z < SMPLX_BRICK; ++z) {
                const unsigned int yz = bfs_window_y(W, zw[z]), xz = bfs_window_x(W, zw[z]);   // (every lane takes part in the shifts)
                const unsigned int hx = tile[(z + 1) * TP + (ty + 1) * TL + xh], hy = tile[(z + 1) * TP + yh * TL + (tx + 1)];
                const unsigned int hxy = tile[(z + 1) * TP + yh * TL + xh];
                fx = fx || can_improve(hx, yz);
                fy = fy || can_improve(hy, xz);
                exy = exy || can_improve(hxy, zw[z]);
            }
            if (on_x && fx) mask |= 1u << (1 * 9 + 1 * 3 + ox);
            if (on_y && fy) mask |= 1u << (1 * 9 + oy * 3 + 1);
            if (on_x && on_y && exy) mask |= 1u << (1 * 9 + oy * 3 + ox);
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const unsigned int vz = side == 0 ? v[0] : v[SMPLX_BRICK - 1];
                const int zs = side == 0 ? 0 : TL - 1, oz = side == 0 ? 0 : 2;
                const unsigned int xw = bfs_window_x(W, vz), yw = bfs_window_y(W, vz), xy = bfs_window_y(W, xw);
                if (can_improve(tile[zs * TP + col], xy)) mask |= 1u << (oz * 9 + 1 * 3 + 1);
                if (on_x && can_improve(tile[zs * TP + (ty + 1) * TL + xh], yw)) mask |= 1u << (oz * 9 + 1 * 3 + ox);
                if (on_y && can_improve(tile[zs * TP + yh * TL + (tx + 1)], xw)) mask |= 1u << (oz * 9 + oy * 3 + 1);
                if (on_x && on_y && can_improve(tile[zs * TP + yh * TL + xh], vz)) mask |= 1u << (oz * 9 + oy * 3 + ox);
            }
        }
        BFS_MARK(5);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mask |= (unsigned int)__shfl_xor((int)mask, off);
        int claimed = -1;
        if (t < 27 && t != 13 && ((mask >> t) & 1u)) {
            const int qx = bxx + t % 3 - 1, qy = byy + (t / 3) % 3 - 1, qz = bzz + t / 9 - 1;
            if (!(qx < 0 || qy < 0 || qz < 0 || qx >= nbx || qy >= nby || qz >= nbz)) {
                const int q = (qz * nby + qy) * nbx + qx;
                if (atomicExch(&queued_next[q], 1) == 0) claimed = q;
            }
        }
        const unsigned long long cm = __ballot(claimed >= 0);
        if (cm != 0) {
            const int shard = (int)(blockIdx.x % SMPLX_BFS_SHARDS);
            int base = 0;
            if (t == 0) base = atomicAdd(&counts_next[32 * shard], __popcll(cm));
            base = __shfl(base, 0);
            if (claimed >= 0) {
                const int pos = base + __popcll(cm & ((1ull << t) - 1ull));
                if (pos < shard_cap) list_next[(size_t)shard * shard_cap + pos] = claimed;
            }
        }
        BFS_MARK(6);
        __syncthreads();   // the tile is reused by the block's next brick
    }
}

#include "search_kernel.h"
