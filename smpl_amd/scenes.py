"""Seeded synthetic inputs for the ARA* state-expansion path.

Nothing here is on the hot path: it builds the *inputs* the path consumes --
a plain-text robot model (replaces URDF + collision-model YAML, which need
urdf/XmlRpc), a motion-primitive file in the reference's own format
(smpl/src/graph/manip_lattice_action_space.cpp:89-103), and a dense squared
distance grid laid out like smpl's Grid3 (x-major, z fastest) with border cells
acting as obstacles (smpl/include/smpl/distance_map/detail/distance_map.hpp:560-606).

The distance field is an exact Euclidean transform (scipy) capped at
ceil(max_dist/res)^2; the reference's incremental propagation
(distance_map.hpp:627-839) is SURVEY row N1 ("next") and is an *input* to the
path, so both the oracle and the HIP engine are always fed the same array.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

DEG = math.pi / 180.0


# ----------------------------------------------------------------------------
# robots
# ----------------------------------------------------------------------------

def arm7_text(prefix: str = "", mount=(0.0, -0.188, 0.80), with_header: bool = True,
              parent: str = "base_link") -> str:
    """7-R arm, PR2-right-arm-like topology: Z,Y,X,Y,X,Y,X axes, two continuous rolls,
    sphere trees of 1,4,7,3,2,2,2,2 leaves (23 leaves), gripper fingers on fixed joints."""
    p = prefix
    L = []
    if with_header:
        L += ["robot arm7", "link base_link"]
    for n in ["shoulder_base", "shoulder_pan_link", "shoulder_lift_link", "upper_arm_link", "elbow_link",
              "forearm_link", "wrist_flex_link", "gripper_palm_link", "l_finger_link", "l_finger_tip_link",
              "r_finger_link", "r_finger_tip_link", "tool_link"]:
        L.append(f"link {p}{n}")
    J = "joint {n} {t} {pa} {ch}  {o[0]} {o[1]} {o[2]}  {r[0]} {r[1]} {r[2]}  {a[0]} {a[1]} {a[2]}  {lo} {hi}"

    def joint(n, t, pa, ch, o=(0, 0, 0), r=(0, 0, 0), a=(0, 0, 1), lo=0.0, hi=0.0):
        pa_name = parent if pa == parent else p + pa
        L.append(J.format(n=p + n, t=t, pa=pa_name, ch=p + ch, o=o, r=r, a=a, lo=lo, hi=hi))

    joint("mount", "fixed", parent, "shoulder_base", o=mount)
    joint("shoulder_pan", "revolute", "shoulder_base", "shoulder_pan_link", a=(0, 0, 1), lo=-2.2, hi=0.7)
    joint("shoulder_lift", "revolute", "shoulder_pan_link", "shoulder_lift_link", o=(0.10, 0, 0), a=(0, 1, 0), lo=-0.5, hi=1.3)
    joint("upper_arm_roll", "revolute", "shoulder_lift_link", "upper_arm_link", a=(1, 0, 0), lo=-3.9, hi=0.8)
    joint("elbow_flex", "revolute", "upper_arm_link", "elbow_link", o=(0.40, 0, 0), a=(0, 1, 0), lo=-2.3, hi=0.0)
    joint("forearm_roll", "continuous", "elbow_link", "forearm_link", a=(1, 0, 0))
    joint("wrist_flex", "revolute", "forearm_link", "wrist_flex_link", o=(0.32, 0, 0), a=(0, 1, 0), lo=-2.1, hi=0.0)
    joint("wrist_roll", "continuous", "wrist_flex_link", "gripper_palm_link", a=(1, 0, 0))
    joint("l_finger", "fixed", "gripper_palm_link", "l_finger_link", o=(0.077, 0.01, 0))
    joint("l_finger_tip", "fixed", "l_finger_link", "l_finger_tip_link", o=(0.09, 0.005, 0))
    joint("r_finger", "fixed", "gripper_palm_link", "r_finger_link", o=(0.077, -0.01, 0))
    joint("r_finger_tip", "fixed", "r_finger_link", "r_finger_tip_link", o=(0.09, -0.005, 0))
    joint("tool", "fixed", "gripper_palm_link", "tool_link", o=(0.18, 0, 0))

    def sph(link, name, x, y, z, r, pr=1):
        L.append(f"sphere {p}{link} {p}{name} {x} {y} {z} {r} {pr}")

    sph("shoulder_pan_link", "sh0", 0.11, 0.0, -0.035, 0.15, 5)
    sph("upper_arm_link", "ua0", 0.17, 0.0, -0.012, 0.105, 4)
    sph("upper_arm_link", "ua1", 0.255, 0.0, -0.03, 0.08, 4)
    sph("upper_arm_link", "ua2", 0.325, 0.0, -0.03, 0.08, 4)
    sph("upper_arm_link", "ua3", 0.40, 0.0, 0.0, 0.10, 2)
    sph("forearm_link", "fa0", 0.125, 0.0, 0.004, 0.075, 2)
    sph("forearm_link", "fa1", 0.195, 0.024, -0.01, 0.055, 3)
    sph("forearm_link", "fa2", 0.195, -0.024, -0.01, 0.055, 3)
    sph("forearm_link", "fa3", 0.26, 0.027, -0.014, 0.05, 3)
    sph("forearm_link", "fa4", 0.26, -0.027, -0.014, 0.05, 3)
    sph("forearm_link", "fa5", 0.315, 0.022, -0.005, 0.05, 3)
    sph("forearm_link", "fa6", 0.315, -0.022, -0.005, 0.05, 3)
    sph("gripper_palm_link", "gr0", 0.07, -0.017, 0.0, 0.04, 2)
    sph("gripper_palm_link", "gr1", 0.07, 0.017, 0.0, 0.04, 2)
    sph("gripper_palm_link", "gr2", 0.09, 0.0, 0.0, 0.04, 2)
    sph("l_finger_link", "lf0", 0.03, 0.024, 0.0, 0.038, 1)
    sph("l_finger_link", "lf1", 0.078, 0.017, 0.0, 0.034, 1)
    sph("r_finger_link", "rf0", 0.03, -0.024, 0.0, 0.038, 1)
    sph("r_finger_link", "rf1", 0.078, -0.017, 0.0, 0.034, 1)
    sph("l_finger_tip_link", "lt0", 0.015, -0.005, 0.0, 0.028, 1)
    sph("l_finger_tip_link", "lt1", 0.034, -0.005, 0.0, 0.02, 1)
    sph("r_finger_tip_link", "rt0", 0.015, 0.005, 0.0, 0.028, 1)
    sph("r_finger_tip_link", "rt1", 0.034, 0.005, 0.0, 0.02, 1)
    return "\n".join(L) + "\n"


_ARM_SPHERE_LINKS = ["shoulder_pan_link", "upper_arm_link", "forearm_link", "gripper_palm_link",
                     "l_finger_link", "r_finger_link", "l_finger_tip_link", "r_finger_tip_link"]
_ARM_JOINTS = ["shoulder_pan", "shoulder_lift", "upper_arm_roll", "elbow_flex", "forearm_roll",
               "wrist_flex", "wrist_roll"]


def _arm_acm(p: str) -> list[str]:
    """Pairs that are never checked (an SRDF-like 'never in collision' list) besides adjacent links."""
    a = []

    def allow(x, y):
        a.append(f"acm {p}{x} {p}{y}")

    allow("shoulder_pan_link", "upper_arm_link")
    allow("upper_arm_link", "forearm_link")
    allow("forearm_link", "gripper_palm_link")
    for f in ["l_finger_link", "r_finger_link", "l_finger_tip_link", "r_finger_tip_link"]:
        allow("forearm_link", f)
    allow("gripper_palm_link", "l_finger_tip_link")
    allow("gripper_palm_link", "r_finger_tip_link")
    allow("l_finger_link", "r_finger_link")
    allow("l_finger_link", "r_finger_tip_link")
    allow("r_finger_link", "l_finger_tip_link")
    allow("l_finger_tip_link", "r_finger_tip_link")
    return a


def arm7_robot() -> str:
    t = arm7_text()
    t += "group arm " + " ".join(_ARM_SPHERE_LINKS) + "\n"
    t += "\n".join(_arm_acm("")) + "\n"
    t += "planning_joints " + " ".join(_ARM_JOINTS) + "\n"
    t += "planning_link tool_link\n"
    return t


def dual_arm14_robot() -> str:
    """Two arm7 chains on a common base (config 5): 14 variables, 46 leaves, inter-arm pairs checked."""
    t = "robot dual_arm14\nlink base_link\n"
    t += arm7_text("r_", mount=(0.0, -0.188, 0.80), with_header=False)
    t += arm7_text("l_", mount=(0.0, 0.188, 0.80), with_header=False)
    t += "group arms " + " ".join(["r_" + l for l in _ARM_SPHERE_LINKS] + ["l_" + l for l in _ARM_SPHERE_LINKS]) + "\n"
    t += "\n".join(_arm_acm("r_") + _arm_acm("l_")) + "\n"
    # the two shoulders sit next to each other on the torso and always overlap
    t += "acm r_shoulder_pan_link l_shoulder_pan_link\n"
    t += "planning_joints " + " ".join(["r_" + j for j in _ARM_JOINTS] + ["l_" + j for j in _ARM_JOINTS]) + "\n"
    t += "planning_link r_tool_link\n"
    return t


MIXED_LIMITS = [(-1.5, 1.5), (0.0, 0.25), (-1.2, 1.2), (-math.pi, math.pi), (-1.0, 1.0)]
MIXED_START = [0.1, 0.05, -0.2, 0.3, 0.2]


def mixed_kinds_robot() -> str:
    """A 5-variable test robot that takes every branch of the joint-transform functions
    (sbpl_collision_checking/src/transform_functions.h:95-258): origins with a rotation (general forms), a revolute
    joint about an axis that is none of X/Y/Z (origin * AngleAxis), a prismatic joint (moves along its LOCAL Z whatever
    the axis says), a continuous joint, identity-origin joints, and a fixed joint with a rotated origin."""
    L = ["robot mixed_kinds", "link base_link"]
    for n in ["pedestal", "turret", "mast", "boom", "roll_link", "wrist", "pad", "tool_link"]:
        L.append(f"link {n}")
    J = "joint {n} {t} {pa} {ch}  {o[0]} {o[1]} {o[2]}  {r[0]} {r[1]} {r[2]}  {a[0]} {a[1]} {a[2]}  {lo} {hi}"

    def joint(n, t, pa, ch, o=(0, 0, 0), r=(0, 0, 0), a=(0, 0, 1), lo=0.0, hi=0.0):
        L.append(J.format(n=n, t=t, pa=pa, ch=ch, o=o, r=r, a=a, lo=lo, hi=hi))

    joint("mount", "fixed", "base_link", "pedestal", o=(0.05, -0.15, 0.55), r=(0.0, 0.0, 0.3))
    joint("pan", "revolute", "pedestal", "turret", o=(0.0, 0.0, 0.10), r=(0.1, 0.2, 0.3), a=(0, 0, 1), lo=-1.5, hi=1.5)
    joint("lift", "prismatic", "turret", "mast", o=(0.05, 0.0, 0.05), r=(0.0, 0.4, 0.0), a=(1, 0, 0), lo=0.0, hi=0.25)
    joint("swing", "revolute", "mast", "boom", o=(0.0, 0.0, 0.20), a=(0.6, 0.0, 0.8), lo=-1.2, hi=1.2)
    joint("roll", "continuous", "boom", "roll_link", o=(0.25, 0.0, 0.0), r=(0.0, -0.3, 0.2), a=(1, 0, 0))
    joint("flex", "revolute", "roll_link", "wrist", o=(0.20, 0.0, 0.0), a=(0, 1, 0), lo=-1.0, hi=1.0)
    joint("pad_mount", "fixed", "wrist", "pad", o=(0.08, 0.0, 0.0), r=(0.5, 0.0, 0.0))
    joint("tool", "fixed", "wrist", "tool_link", o=(0.15, 0.0, 0.0))

    def sph(link, name, x, y, z, r, pr=1):
        L.append(f"sphere {link} {name} {x} {y} {z} {r} {pr}")

    sph("turret", "tu0", 0.0, 0.0, 0.05, 0.10, 3)
    sph("mast", "ma0", 0.0, 0.0, 0.08, 0.07, 3)
    sph("mast", "ma1", 0.0, 0.0, 0.18, 0.06, 3)
    sph("boom", "bo0", 0.08, 0.0, 0.0, 0.07, 2)
    sph("boom", "bo1", 0.18, 0.0, 0.0, 0.06, 2)
    sph("roll_link", "ro0", 0.07, 0.0, 0.0, 0.06, 2)
    sph("roll_link", "ro1", 0.15, 0.01, 0.0, 0.05, 2)
    sph("wrist", "wr0", 0.05, 0.0, 0.0, 0.05, 1)
    sph("pad", "pa0", 0.03, 0.02, 0.0, 0.035, 1)
    sph("pad", "pa1", 0.03, -0.02, 0.0, 0.035, 1)
    t = "\n".join(L) + "\n"
    t += "group arm turret mast boom roll_link wrist pad\n"
    t += "acm mast roll_link\nacm boom wrist\nacm roll_link pad\n"
    t += "planning_joints pan lift swing roll flex\n"
    t += "planning_link tool_link\n"
    return t


def config_mixed(n: int = 64, seed: int = 11, nboxes: int = 5) -> "Config":
    """The mixed-kinds robot in a small cluttered scene; the prismatic variable is discretised in metres."""
    res = 0.04
    size = n * res
    origin = (0.4 - 0.5 * size, -0.5 * size, 0.8 - 0.5 * size)
    rng = np.random.default_rng(seed)
    clear = [(0.05, -0.15, 0.6), (0.1, -0.15, 0.9), (0.3, -0.1, 1.0), (0.5, 0.0, 1.0)]
    boxes = [TABLETOP] + random_boxes(rng, nboxes, origin, (n, n, n), res, clear, 0.30)
    grid = build_grid(origin, (n, n, n), res, 0.4, boxes)
    p = PlanningParams([2 * DEG, 0.01, 2 * DEG, 2 * DEG, 2 * DEG], eps0=5.0, bfs_radius=0.04, cost_per_cell=500)
    goal = [MIXED_START[0] + 20 * 2 * DEG, MIXED_START[1] + 8 * 0.01, MIXED_START[2] - 12 * 2 * DEG,
            MIXED_START[3] + 8 * 2 * DEG, MIXED_START[4] - 8 * 2 * DEG]
    return Config("mixed", mixed_kinds_robot(), mprim_text(5, range(3), range(5), long_cells=4, short_cells=2), grid, p,
                  list(MIXED_START), goal, [3.0 * DEG, 0.02, 3.0 * DEG, 3.0 * DEG, 3.0 * DEG], boxes)


def mprim_text(nvars: int, long_joints, short_joints, long_cells: int = 7, short_cells: int = 4) -> str:
    """Upstream .mprim layout (smpl_test/config/pr2.mprim has 4 long rows of 7 and 7 short rows of 4)."""
    rows = []
    for j in long_joints:
        r = [0] * nvars
        r[j] = long_cells
        rows.append(r)
    for j in short_joints:
        r = [0] * nvars
        r[j] = short_cells
        rows.append(r)
    out = [f"Motion_Primitives(degrees): {len(rows)} {nvars} {len(short_joints)}"]
    out += [" ".join(str(v) for v in r) for r in rows]
    return "\n".join(out) + "\n"


# ----------------------------------------------------------------------------
# scenes / grids
# ----------------------------------------------------------------------------

@dataclass
class Grid:
    origin: tuple
    dims: tuple          # interior cells (nx, ny, nz)
    res: float
    max_dist: float
    d2: np.ndarray       # int32 [nx, ny, nz], squared cell distance to nearest obstacle/border, capped

    @property
    def dmax_int(self) -> int:
        return int(math.ceil(self.max_dist * (1.0 / self.res)))


def world_to_grid(origin, res, w):
    """smpl/include/smpl/distance_map/detail/distance_map.hpp:520-527 (vectorised)."""
    inv = 1.0 / res
    return ((inv * (np.asarray(w, dtype=np.float64) - (np.asarray(origin) - res)) + 0.5).astype(np.int64) - 1)


def box_cells(origin, res, dims, center, size):
    """Cells whose centres fall inside an axis-aligned box (minimal restatement of
    smpl/src/geometry/voxelize.cpp:673-735 VoxelizeBox for scene building only)."""
    lo = np.asarray(center) - 0.5 * np.asarray(size)
    hi = np.asarray(center) + 0.5 * np.asarray(size)
    clo = np.maximum(world_to_grid(origin, res, lo), 0)
    chi = np.minimum(world_to_grid(origin, res, hi), np.asarray(dims) - 1)
    return clo, chi


def build_grid(origin, dims, res, max_dist, boxes) -> Grid:
    from scipy import ndimage

    nx, ny, nz = dims
    occ = np.zeros((nx + 2, ny + 2, nz + 2), dtype=bool)
    occ[0, :, :] = occ[-1, :, :] = True
    occ[:, 0, :] = occ[:, -1, :] = True
    occ[:, :, 0] = occ[:, :, -1] = True
    for (c, s) in boxes:
        clo, chi = box_cells(origin, res, dims, c, s)
        if np.all(chi >= clo):
            occ[clo[0] + 1:chi[0] + 2, clo[1] + 1:chi[1] + 2, clo[2] + 1:chi[2] + 2] = True
    dmax = int(math.ceil(max_dist * (1.0 / res)))
    # exact squared distances from the feature transform (integer arithmetic)
    idx = ndimage.distance_transform_edt(~occ, return_distances=False, return_indices=True)
    d2 = np.zeros(occ.shape, dtype=np.int64)
    for a in range(3):
        g = np.arange(occ.shape[a]).reshape([-1 if i == a else 1 for i in range(3)])
        d = idx[a].astype(np.int64) - g
        d2 += d * d
    del idx
    d2 = np.minimum(d2, dmax * dmax).astype(np.int32)
    return Grid(tuple(origin), tuple(dims), res, max_dist, np.ascontiguousarray(d2[1:-1, 1:-1, 1:-1]))


def random_boxes(rng: np.random.Generator, n, origin, dims, res, keep_clear, clear_radius, edge=(0.05, 0.30)):
    """n random boxes, rejecting those within clear_radius of any keep_clear point."""
    size = np.asarray(dims) * res
    boxes = []
    keep_clear = np.asarray(keep_clear, dtype=np.float64).reshape(-1, 3)
    while len(boxes) < n:
        e = rng.uniform(edge[0], edge[1], size=3)
        c = np.asarray(origin) + rng.uniform(0.0, 1.0, size=3) * size
        half_diag = 0.5 * float(np.linalg.norm(e))
        if keep_clear.size and np.min(np.linalg.norm(keep_clear - c, axis=1)) < clear_radius + half_diag:
            continue
        boxes.append((tuple(c), tuple(e)))
    return boxes


@dataclass
class PlanningParams:
    resolutions: list
    bfs_radius: float = 0.02          # smpl_test/src/call_planner.cpp:1715
    cost_per_cell: int = 250
    use_short: bool = True            # smpl_test/config/pr2_right_arm.yaml use_short_dist_mprims
    short_thresh: float = 0.4
    use_xyzrpy_snap: bool = True
    xyzrpy_thresh: float = 0.04
    # [FORK] manip_lattice_action_space.cpp:590-599 rotates delta[0], delta[1] by state[3] in every
    # applyMotionPrimitive: that is the reference's behaviour and the default here; False = upstream smpl
    xy_rotate_by_var3: bool = True
    use_long_and_short: bool = False
    eps0: float = 5.0
    eps_final: float = 1.0
    eps_delta: float = 1.0


@dataclass
class Config:
    name: str
    robot_text: str
    mprim: str
    grid: Grid
    params: PlanningParams
    start: list
    goal: list
    goal_tol: list = field(default_factory=list)
    boxes: list = field(default_factory=list)


ARM7_START = [0.0, 0.0, 0.0, -1.1356, 0.0, -1.05, 0.0]   # smpl_test/experiments/pr2_goal.yaml
# lattice-reachable goal: joints 0-3 any cell, joints 4-6 a multiple of 4 cells from the start
ARM7_GOAL_CELLS = [-88, 4, 29, -22, -32, -52, 40]
TABLETOP = ((0.55, 0.0, 0.6), (0.4, 1.5, 0.02))             # smpl_test/env/tabletop.env


def _arm7_goal():
    return [ARM7_START[i] + ARM7_GOAL_CELLS[i] * DEG for i in range(7)]


def config1(n: int = 128) -> Config:
    """SURVEY 8d cfg 1: 7-DOF arm, 128^3 @ 0.02 m, tabletop, eps 100 -> 1 (CPU plumbing case)."""
    origin = (-0.75, -1.28, 0.0)
    grid = build_grid(origin, (n, n, n), 0.02, 0.4, [TABLETOP])
    p = PlanningParams([DEG] * 7, eps0=100.0)
    return Config("cfg1", arm7_robot(), mprim_text(7, range(4), range(7)), grid, p, list(ARM7_START), _arm7_goal(),
                  [3.0 * DEG] * 7, [TABLETOP])


def config2(n: int = 256, nboxes: int = 64, seed: int = 2) -> Config:
    """SURVEY 8d cfg 2: same arm, 256^3 @ 0.02 m, tabletop + 64 random boxes (seed 2), eps 5 -> 1."""
    res = 0.02
    size = n * res
    origin = (0.5 - 0.5 * size, -0.5 * size, 0.8 - 0.25 * size)
    rng = np.random.default_rng(seed)
    # keep the arm's swept volume between start and goal roughly clear
    clear = [(0.0, -0.188, 0.8), (0.45, -0.188, 0.8), (0.6, -0.3, 0.9), (0.5, -0.6, 0.95), (0.3, -0.7, 1.0)]
    boxes = [TABLETOP] + random_boxes(rng, nboxes, origin, (n, n, n), res, clear, 0.30)
    grid = build_grid(origin, (n, n, n), res, 0.4, boxes)
    p = PlanningParams([DEG] * 7, eps0=5.0)
    return Config("cfg2", arm7_robot(), mprim_text(7, range(4), range(7)), grid, p, list(ARM7_START), _arm7_goal(),
                  [3.0 * DEG] * 7, boxes)


def config3(n: int = 150, nclutter: int = 20, seed: int = 3) -> Config:
    """SURVEY 8d cfg 3: PR2-right-arm-like model as data, 150^3 @ 0.02 m, origin (-0.75,-1.5,0), max_dist 1.8
    (smpl_test/src/call_planner.cpp:1587-1594), cluttered tabletop (seed 3); base and torso enter the grid as
    static boxes (in the reference they are voxelised out-of-group links).  The grid is not a multiple of the
    4-cell brick and squared distances reach 8100."""
    res = 0.02
    origin = (-0.75, -1.5, 0.0)
    rng = np.random.default_rng(seed)
    boxes = [TABLETOP, ((-0.05, 0.0, 0.18), (0.65, 0.65, 0.36)), ((-0.32, 0.0, 0.62), (0.3, 0.35, 0.5))]
    while len(boxes) < 3 + nclutter:       # clutter standing on the table
        e = rng.uniform(0.04, 0.12, size=3)
        c = (0.35 + rng.uniform(0.05, 0.35), rng.uniform(-0.7, 0.7), 0.61 + 0.5 * e[2])
        if abs(c[1] + 0.188) < 0.25 and c[0] < 0.6:
            continue                       # keep the corridor in front of the shoulder free
        boxes.append((tuple(c), tuple(e)))
    grid = build_grid(origin, (n, n, n), res, 1.8, boxes)
    p = PlanningParams([DEG] * 7, eps0=100.0)     # call_planner.cpp:1727-1729
    goal = [ARM7_START[i] + c * DEG for i, c in enumerate([-49, 7, 21, -14, -8, -12, 16])]
    return Config("cfg3", arm7_robot(), mprim_text(7, range(4), range(7)), grid, p, list(ARM7_START), goal,
                  [3.0 * DEG] * 7, boxes)


PR2_RIGHT_ARM_JOINTS = ["r_shoulder_pan_joint", "r_shoulder_lift_joint", "r_upper_arm_roll_joint", "r_elbow_flex_joint",
                        "r_forearm_roll_joint", "r_wrist_flex_joint", "r_wrist_roll_joint"]   # smpl_test/launch/goal_pr2.launch:16-23


def pr2_right_arm_text(collision_yaml: str, urdf_xml: str, allowed_pairs=()) -> str:
    """SURVEY 8d cfg 3, "PR2 right arm as data": the plain-text model built from the reference's own collision-model file
    (sbpl_collision_checking_test/config/collision_model_pr2.yaml: 23 leaf spheres on 8 links of group `right_arm`,
    chains and nested groups resolved as robot_collision_model.cpp:517-623 does) and a URDF subset holding the arm's
    kinematics (the reference tree has no PR2 URDF: goal_pr2.launch takes it from pr2_description).  Planning joints and
    planning link as goal_pr2.launch:14-29; `allowed_pairs`: link pairs never checked against each other (the right-arm rows
    of the demo's allowed-collision matrix, call_planner.cpp:441-1526)."""
    from . import formats
    links = formats.group_links_from_collision_yaml(collision_yaml, "right_arm", urdf_xml)
    spheres = formats.sphere_lines_from_collision_yaml(collision_yaml, links=links)
    return formats.urdf_to_robot_text(urdf_xml, "right_arm", links, PR2_RIGHT_ARM_JOINTS, "r_gripper_palm_link", spheres,
                                      [tuple(p) for p in allowed_pairs])


def config3_pr2(collision_yaml: str, urdf_xml: str, allowed_pairs=(), n: int = 150) -> Config:
    """cfg 3 with the PR2 right arm built from data files (pr2_right_arm_text) in the cfg-3 scene."""
    import dataclasses
    base = config3(n)
    goal = [ARM7_START[i] + c * DEG for i, c in enumerate([-49, 7, 21, -14, -8, -12, 16])]
    return dataclasses.replace(base, name="cfg3_pr2", robot_text=pr2_right_arm_text(collision_yaml, urdf_xml, allowed_pairs), goal=goal)


def config_small(n: int = 64, seed: int = 7, nboxes: int = 6) -> Config:
    """Small test scene (64^3 @ 0.04 m): the oracle finishes a full ARA* query in well under a second."""
    res = 0.04
    size = n * res
    origin = (0.5 - 0.5 * size, -0.5 * size, 0.8 - 0.5 * size)
    rng = np.random.default_rng(seed)
    clear = [(0.0, -0.188, 0.8), (0.45, -0.188, 0.8), (0.6, -0.3, 0.9), (0.5, -0.6, 0.95), (0.3, -0.7, 1.0)]
    boxes = [TABLETOP] + random_boxes(rng, nboxes, origin, (n, n, n), res, clear, 0.30)
    grid = build_grid(origin, (n, n, n), res, 0.4, boxes)
    p = PlanningParams([DEG] * 7, eps0=5.0, bfs_radius=0.04, cost_per_cell=500)
    goal = [ARM7_START[i] + c * DEG for i, c in enumerate([-49, 7, 21, -14, -8, -12, 16])]
    return Config("small", arm7_robot(), mprim_text(7, range(4), range(7)), grid, p, list(ARM7_START), goal,
                  [3.0 * DEG] * 7, boxes)


def config5(n: int = 512, nboxes: int = 128, seed: int = 5, res: float = 0.01) -> Config:
    """SURVEY 8d cfg 5: 14-DOF dual arm, 512^3 @ 0.01 m, tabletop + 128 boxes, eps 10 -> 1.
    (n=64, res=0.08 gives the same scene at a size the oracle handles in tests.)"""
    size = n * res
    origin = (0.5 - 0.5 * size, -0.5 * size, 0.8 - 0.4 * size)
    rng = np.random.default_rng(seed)
    clear = [(0.0, -0.188, 0.8), (0.0, 0.188, 0.8), (0.45, -0.188, 0.8), (0.45, 0.188, 0.8), (0.6, -0.3, 0.9),
             (0.6, 0.3, 0.9), (0.5, -0.6, 0.95), (0.5, 0.6, 0.95)]
    boxes = [TABLETOP] + random_boxes(rng, nboxes, origin, (n, n, n), res, clear, 0.30, edge=(0.05, 0.25))
    grid = build_grid(origin, (n, n, n), res, 0.4, boxes)
    p = PlanningParams([DEG] * 14, eps0=10.0, bfs_radius=max(0.02, res))
    start = list(ARM7_START) + list(ARM7_START)
    start[7] = 0.0
    goal = _arm7_goal() + [ARM7_START[i] - ARM7_GOAL_CELLS[i] * DEG * (1 if i in (1, 3, 5) else -1) for i in range(7)]
    # mirror the left arm's pan/roll so it reaches to its own side
    goal[7] = ARM7_START[0] + 28 * DEG
    goal[9] = ARM7_START[2] + 35 * DEG
    goal[11] = ARM7_START[4] - 8 * DEG
    goal[13] = ARM7_START[6] + 16 * DEG
    return Config("cfg5", dual_arm14_robot(), mprim_text(14, range(14), range(14)), grid, p, start, goal,
                  [3.0 * DEG] * 14, boxes)


def with_extra_spheres(robot_text: str, n: int, links=("shoulder_pan_link", "forearm_link")) -> str:
    """The robot with n more small spheres in a row on each of two links that form a checked pair: deeper sphere trees
    (tests of the traversal-stack sizing)."""
    lines = robot_text.split("\n")
    i = max(k for k, l in enumerate(lines) if l.startswith("sphere "))
    extra = [f"sphere {link} x{j}_{k} {0.02 * k:.6f} 0.0 0.0 0.02 1" for j, link in enumerate(links) for k in range(n)]
    return "\n".join(lines[:i + 1] + extra + lines[i + 1:])


def random_states(cfg_limits, n: int, seed: int = 12345) -> np.ndarray:
    """n joint states uniform within limits (scheme of
    sbpl_collision_checking_test/src/benchmark_cc.cpp:280-301); continuous joints in [-pi, pi]."""
    rng = np.random.default_rng(seed)
    lo = np.array([l for (l, h) in cfg_limits])
    hi = np.array([h for (l, h) in cfg_limits])
    return lo + rng.uniform(size=(n, len(cfg_limits))) * (hi - lo)


ARM7_LIMITS = [(-2.2, 0.7), (-0.5, 1.3), (-3.9, 0.8), (-2.3, 0.0), (-math.pi, math.pi), (-2.1, 0.0),
               (-math.pi, math.pi)]


# ----------------------------------------------------------------------------
# BASELINE config 4: 1024 independent (start, goal) queries on the config-2 scene, 128 per GPU
# ----------------------------------------------------------------------------

def config4_candidates(n: int = 1024, seed: int = 4, oversample: int = 3):
    """Candidate (start, goal) pairs of SURVEY 8d cfg 4 (seed 4): whole-cell offsets from the cfg-2 start and goal
    (1 degree cells; joints 4-6 in multiples of 4 cells, the short primitives' step).  Returns two arrays of
    oversample*n rows; the caller keeps the first n pairs whose two ends its collision checker accepts
    (config4_queries) -- the engine in bench.py and the tests, so that the list is a function of seed and scene only."""
    rng = np.random.default_rng(seed)
    m = n * oversample
    ds = rng.integers(-20, 21, size=(m, 7))
    dg = rng.integers(-12, 13, size=(m, 7))
    ds[:, 4:] = 4 * rng.integers(-4, 5, size=(m, 3))
    dg[:, 4:] = 4 * rng.integers(-3, 4, size=(m, 3))
    starts = np.asarray(ARM7_START)[None, :] + ds * DEG
    goals = np.asarray(_arm7_goal())[None, :] + dg * DEG
    lo = np.array([l for (l, h) in ARM7_LIMITS]); hi = np.array([h for (l, h) in ARM7_LIMITS])
    cont = np.array([False, False, False, False, True, False, True])
    inside = lambda q: np.all((q >= lo) & (q <= hi) | cont, axis=1)   # noqa: E731
    keep = inside(starts) & inside(goals)
    return starts[keep], goals[keep]


def config4_queries(starts, goals, start_ok, goal_ok, n: int = 1024):
    """First n candidate pairs with both ends valid; query i belongs to rank i // 128."""
    ok = np.asarray(start_ok, bool) & np.asarray(goal_ok, bool)
    idx = np.nonzero(ok)[0][:n]
    if idx.shape[0] < n:
        raise ValueError(f"only {idx.shape[0]} valid pairs among {ok.shape[0]} candidates")
    return starts[idx], goals[idx]


def shard_range(rank: int, world: int, total: int = 1024, per_rank: int = 128):
    """Queries [first, last) of rank `rank`: 128 per GPU (BASELINE config 4); with fewer than total/per_rank ranks the
    job simply covers a prefix of the list (weak scaling: per-GPU work is fixed)."""
    first = min(rank * per_rank, total)
    return first, min(rank * per_rank + per_rank, total) if rank * per_rank < total else first


# ----------------------------------------------------------------------------
# K2 micro-benchmark inputs: the same DISTRIBUTION as sbpl_collision_checking_test/src/benchmark_cc.cpp:280-301 (one
# uniform_real_distribution per variable), with the engine and seed SURVEY 8d names (std::mt19937_64, 12345).  The reference
# binary draws from a default-seeded std::default_random_engine (benchmark_cc.cpp:165), so its exact states differ, and no
# reference fixture pins valid_fraction or the lookup tally: K2 parity is against the oracle only.
# ----------------------------------------------------------------------------

class MT19937_64:
    """std::mt19937_64 (the C++ standard's 64-bit Mersenne twister), block-vectorised in numpy.  Pinned by the
    standard's own known answer: the 10000th output of a default-seeded (5489) engine is 9981545732273789042."""
    NN, MM = 312, 156

    def __init__(self, seed: int = 5489):
        mt = np.zeros(self.NN, np.uint64)
        x = seed & 0xFFFFFFFFFFFFFFFF
        mt[0] = x
        for i in range(1, self.NN):
            x = (6364136223846793005 * (x ^ (x >> 62)) + i) & 0xFFFFFFFFFFFFFFFF
            mt[i] = x
        self.mt = mt
        self.buf = np.zeros(0, np.uint64)

    def _twist(self):
        mt = self.mt
        UM, LM, A = np.uint64(0xFFFFFFFF80000000), np.uint64(0x7FFFFFFF), np.uint64(0xB5026F5AA96619E9)
        one = np.uint64(1)

        def mix(cur, nxt, far):
            x = (cur & UM) | (nxt & LM)
            return far ^ (x >> one) ^ np.where((x & one) != 0, A, np.uint64(0))
        # element i needs old mt[i+1] and mt[i+156] (new values once i+156 wraps): three dependency-free chunks
        n, m = self.NN, self.MM
        mt[0:n - m] = mix(mt[0:n - m], mt[1:n - m + 1], mt[m:n])                   # i in [0, 156): far = old
        mt[n - m:n - 1] = mix(mt[n - m:n - 1], mt[n - m + 1:n], mt[0:m - 1])      # i in [156, 311): far = new mt[i-156]
        mt[n - 1:n] = mix(mt[n - 1:n], mt[0:1], mt[m - 1:m])
        y = mt.copy()
        y ^= (y >> np.uint64(29)) & np.uint64(0x5555555555555555)
        y ^= (y << np.uint64(17)) & np.uint64(0x71D67FFFEDA60000)
        y ^= (y << np.uint64(37)) & np.uint64(0xFFF7EEE000000000)
        y ^= y >> np.uint64(43)
        return y

    def raw(self, n: int) -> np.ndarray:
        out = [self.buf]
        have = self.buf.shape[0]
        while have < n:
            out.append(self._twist())
            have += self.NN
        allv = np.concatenate(out)
        self.buf = allv[n:]
        return allv[:n]

    def uniform(self, lo, hi, n: int) -> np.ndarray:
        """n draws of std::uniform_real_distribution<double>(lo, hi) as libstdc++ computes them: generate_canonical
        takes one 64-bit output, divides by 2^64 (a value that rounds to 1.0 becomes the double below 1.0)."""
        u = self.raw(n).astype(np.float64) * (1.0 / 18446744073709551616.0)
        u = np.where(u >= 1.0, np.nextafter(1.0, 0.0), u)
        return u * (hi - lo) + lo


def benchmark_states(limits, n: int = 1 << 20, seed: int = 12345) -> np.ndarray:
    """n joint states q ~ U[limits], one uniform_real_distribution per variable drawn in variable order for each state
    (same distribution as benchmark_cc.cpp:280-301), SURVEY-specified engine and seed: std::mt19937_64, 12345 (SURVEY 8d K2
    micro-benchmark).  Not the states the reference binary would check (it seeds differently); parity is against the oracle."""
    g = MT19937_64(seed)
    nv = len(limits)
    u = g.raw(n * nv).astype(np.float64) * (1.0 / 18446744073709551616.0)
    u = np.where(u >= 1.0, np.nextafter(1.0, 0.0), u).reshape(n, nv)
    lo = np.array([l for (l, h) in limits]); hi = np.array([h for (l, h) in limits])
    return u * (hi - lo) + lo
