"""Known-answer tests of the CPU oracle, hand-derived from the cited reference lines (SURVEY.md 8c list).
The reference's own tests hold no numeric vectors for this path, so these are the pin ("parity unpinned"
by the reference; see oracle/smpl_oracle.hpp header)."""
import math

import numpy as np
import pytest

from oracle_binding import Oracle
from smpl_amd import scenes

DEG = scenes.DEG


@pytest.fixture(scope="module")
def o(small_cfg):
    x = Oracle(small_cfg)
    x.set_goal_joint(small_cfg.goal, small_cfg.goal_tol)
    return x


def test_sincos_within_one_ulp_of_libm(o):
    rng = np.random.default_rng(0)
    for x in np.concatenate([rng.uniform(-20, 20, 20000), [0.0, math.pi / 2, -math.pi, 1e-9, 7 * DEG]]):
        s, c = o.sincos(float(x))
        # <= 1 ulp, or 1e-19 absolute next to a zero of the function (3-term Cody-Waite cancellation)
        assert abs(s - math.sin(x)) <= max(math.ulp(math.sin(x)), 1e-19)
        assert abs(c - math.cos(x)) <= max(math.ulp(math.cos(x)), 1e-19)
    assert o.sincos(0.0) == (0.0, 1.0)


def test_normalize_angle_branches(o):
    # smpl/include/smpl/angles.h:45-62
    f = o.lib.orc_normalize_angle
    assert f(0.5) == 0.5
    assert f(math.pi) == math.pi                      # not > pi: unchanged
    assert f(-math.pi) == -math.pi
    assert f(3.5) == 3.5 - 2 * math.pi
    assert f(-3.5) == -3.5 + 2 * math.pi
    assert f(7.0) == math.fmod(7.0, 2 * math.pi)      # |a| > 2pi: fmod first, result 0.7168 stays


def test_state_to_coord_bin_edges_and_wrap(o):
    # manip_lattice.cpp:1263-1289; variable 4 is continuous (360 cells of 1 degree), variable 0 bounded from -2.2
    vals, deltas = o.model()["coord_vals"], o.model()["coord_deltas"]
    assert vals[4] == 360 and deltas[4] == 2 * math.pi / 360
    q = np.array(scenes.ARM7_START, dtype=float)
    q[4] = 0.0
    assert o.state_to_coord(q)[4] == 0
    q[4] = 0.49 * deltas[4]
    assert o.state_to_coord(q)[4] == 0
    q[4] = 0.51 * deltas[4]
    assert o.state_to_coord(q)[4] == 1
    q[4] = 2 * math.pi - 0.49 * deltas[4]             # last half bin wraps to 0 (coord == vals -> 0)
    assert o.state_to_coord(q)[4] == 0
    q[4] = -0.6 * deltas[4]                           # negative angles are normalised positive first
    assert o.state_to_coord(q)[4] == 359
    # bounded: (int)((q - min)/delta + 0.5)
    span = 0.7 + 2.2
    assert vals[0] == round(span / DEG) and deltas[0] == span / vals[0]
    q[0] = -2.2
    assert o.state_to_coord(q)[0] == 0
    q[0] = -2.2 + 10.49 * deltas[0]
    assert o.state_to_coord(q)[0] == 10
    q[0] = -2.2 + 10.51 * deltas[0]
    assert o.state_to_coord(q)[0] == 11


def test_joint_limits_kdl_semantics(o):
    # kdl_robot_model.cpp:173-189, 210-235
    q = np.array(scenes.ARM7_START, dtype=float)
    assert o.check_joint_limits(q)
    q2 = q.copy(); q2[3] = 0.01                        # elbow upper limit 0.0
    assert not o.check_joint_limits(q2)
    q2 = q.copy(); q2[3] = -2.31
    assert not o.check_joint_limits(q2)
    q2 = q.copy(); q2[4] = 17.0                        # continuous: any value passes
    assert o.check_joint_limits(q2)
    q2 = q.copy(); q2[0] = 0.6 - 2 * math.pi           # a full turn below the range is folded back in
    assert o.check_joint_limits(q2)


def test_world_to_grid_cell_boundaries(o, small_cfg):
    # distance_map.hpp:520-527: (int)(inv_res*(w - (origin - res)) + 0.5) - 1
    g = small_cfg.grid
    ox, oy, oz = g.origin
    assert o.world_to_grid(ox, oy, oz).tolist() == [0, 0, 0]
    assert o.world_to_grid(ox + 0.49 * g.res, oy, oz).tolist() == [0, 0, 0]
    assert o.world_to_grid(ox + 0.51 * g.res, oy, oz).tolist() == [1, 0, 0]
    assert o.world_to_grid(ox - 0.51 * g.res, oy, oz).tolist()[0] == -1
    # out of bounds -> distance 0 (distance_map.hpp:292-296)
    assert o.grid_sqdist(ox - 1.0, oy, oz) == 0.0
    # in bounds -> (res * sqrt(d2))^2, the reference's rounding (distance_map_interface.h:113-114)
    c = o.world_to_grid(0.2, -0.2, 1.0)
    d2 = int(g.d2[c[0], c[1], c[2]])
    d = g.res * math.sqrt(d2)
    assert o.grid_sqdist(0.2, -0.2, 1.0) == d * d


def test_waypoint_count_single_joint_move(o):
    # robot_motion_collision_model.h:173-181,352-366: W = max(2, ceil(k*|dq|/0.05) + 1); 0 if no motion
    m = o.model()
    k_pan = m["k"][1]                                  # joint 1 in file order = shoulder_pan
    a = np.array(scenes.ARM7_START, dtype=float)
    assert o.waypoint_count(a, a) == 0
    b = a.copy(); b[0] += 7 * DEG
    assert o.waypoint_count(a, b) == max(2, math.ceil(k_pan * abs(b[0] - a[0]) / 0.05) + 1)
    b = a.copy(); b[0] += 1e-6
    assert o.waypoint_count(a, b) == 2                  # ceil(tiny)+1 = 2
    # continuous joint: shortest angular distance (robot_motion_collision_model.cpp:388-391)
    k_wr = m["k"][7]
    b = a.copy(); b[6] += 2 * math.pi - 4 * DEG
    w = o.waypoint_count(a, b)
    assert w == max(2, math.ceil(k_wr * abs(o.lib.orc_normalize_angle(b[6] - a[6])) / 0.05) + 1)


def test_mprim_gating_truth_table(o, small_cfg):
    # manip_lattice_action_space.cpp:662-691 with short primitives enabled, thresh 0.4
    far = np.array(small_cfg.start, dtype=float); far[0] = 0.6; far[1] = -0.3   # tool far from the goal
    near = np.array(small_cfg.goal, dtype=float); near[0] += 4 * DEG
    pf, pn = o.planning_fk(far), o.planning_fk(near)
    assert o.metric_goal_distance(*pf) > 0.4 >= o.metric_goal_distance(*pn)
    ef, en = o.eval_state(far)["flags"], o.eval_state(near)["flags"]
    long_idx, short_idx = range(3, 11), range(11, 25)
    assert all(ef[i] != 0x10 for i in long_idx) and all(ef[i] == 0x10 for i in short_idx)
    assert all(en[i] == 0x10 for i in long_idx) and all(en[i] != 0x10 for i in short_idx)
    assert ef[0] == ef[1] == 0x10 and en[0] == en[1] == 0x10          # IK snaps never fire (no IK)
    assert ef[2] == 0x10 and en[2] != 0x10                              # goal-as-IK snap only near the goal


def test_bfs_matches_bruteforce_and_sentinels(o, small_cfg):
    # bfs3d.cpp:507-547: 26-connected hop counts; WALL 0x7FFFFFFF; unreachable stays -1
    from collections import deque
    g = o.bfs_grid()                                   # [z][y][x], padded
    assert g[0, 0, 0] == 0x7FFFFFFF
    p = o.goal_pose()
    c = o.world_to_grid(*p)
    assert g[c[2] + 1, c[1] + 1, c[0] + 1] == 0
    wall = g == 0x7FFFFFFF
    dist = np.full(g.shape, -1, np.int32)
    dist[wall] = 0x7FFFFFFF
    src = (c[2] + 1, c[1] + 1, c[0] + 1)
    dist[src] = 0
    dq = deque([src])
    offs = [(a, b, d) for a in (-1, 0, 1) for b in (-1, 0, 1) for d in (-1, 0, 1) if (a, b, d) != (0, 0, 0)]
    while dq:
        z, y, x = dq.popleft()
        for a, b, d in offs:
            n = (z + a, y + b, x + d)
            if dist[n] == -1:
                dist[n] = dist[z, y, x] + 1
                dq.append(n)
    assert np.array_equal(dist, g)
    # getBfsCostToGoal (bfs_heuristic.cpp:355-366): out of bounds and walls -> 32767
    q = np.array(small_cfg.start, dtype=float)
    assert o.heuristic_q(q) == small_cfg.params.cost_per_cell * int(g[tuple(o.world_to_grid(*o.planning_fk(q))[::-1] + 1)])


def test_tree_traversal_orders_agree_on_booleans(o):
    Q = scenes.random_states(scenes.ARM7_LIMITS, 3000, 21)
    o.set_order(chain=False)
    ref = [o.state_valid(q)[0] for q in Q]
    o.set_order(chain=True)
    ch = [o.state_valid(q)[0] for q in Q]
    o.set_order(chain=False)
    assert ref == ch and 0.05 < np.mean(ref) < 0.95


def test_tree_check_versus_bruteforce_leaves(o, small_cfg):
    """collision_operations.h:105-164 fails only at a leaf that fails its own test, so 'tree says collision'
    implies 'some leaf collides'.  The converse does NOT hold exactly: an inner bounding sphere is tested at the
    cell of ITS centre (distance_map.hpp:520-527), and cell quantisation lets it clear the grid while a leaf
    under it does not.  The tree traversal -- not the leaf set -- is therefore the semantics to match
    (SURVEY a19's 'only the set of leaves matters' is an approximation), which is why the oracle and the HIP
    kernels both walk the reference's trees, built with the reference's partition order."""
    m = o.model()
    leaves = np.where(m["left"] < 0)[0]
    first = m["tree_first"]
    tree_of = np.zeros(len(m["left"]), int)
    for t in range(len(first) - 1):
        tree_of[first[t]:first[t + 1]] = t
    pairs = {tuple(p) for p in m["pairs"].tolist()}
    Q = scenes.random_states(scenes.ARM7_LIMITS, 1500, 22)
    hidden = 0
    for q in Q:
        pos = o.sphere_positions(q, len(m["left"]))
        brute_ok = True
        for i in leaves:
            r = m["xyzr"][i, 3]
            if not (o.grid_sqdist(*pos[i]) >= r * r):
                brute_ok = False
        for i in leaves:
            for j in leaves:
                if (tree_of[i], tree_of[j]) in pairs:
                    d = pos[j] - pos[i]
                    rr = m["xyzr"][i, 3] + m["xyzr"][j, 3]
                    if not ((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2] > rr * rr):
                        brute_ok = False
        tree_ok = o.state_valid(q)[0]
        if brute_ok:
            assert tree_ok                      # no false collisions
        if tree_ok and not brute_ok:
            hidden += 1                         # a colliding leaf hidden behind a clearing bounding sphere
    assert hidden < 0.02 * len(Q)


def test_arastar_key_truncation_and_bounded_plan_is_deterministic(small_cfg):
    # arastar.cpp:579-582: f = g + (unsigned)(eps*h); two identical bounded runs give identical logs
    runs = []
    for _ in range(2):
        x = Oracle(small_cfg)
        x.set_goal_joint(small_cfg.goal, small_cfg.goal_tol)
        assert x.set_start(small_cfg.start) == 1       # id 0 = goal (manip_lattice.cpp:122), start = 1
        x.search_params(5.0, 1.0, 1.0, True, True, 4000, 3000)
        runs.append(x.plan())
    assert runs[0]["ok"] == 1 and runs[0]["cost"] == runs[1]["cost"] > 0
    assert np.array_equal(runs[0]["expansion_log"], runs[1]["expansion_log"])
    assert runs[0]["path"][0] == 1 and runs[0]["path"][-1] == 0
    assert runs[0]["cost"] % 500 == 0                   # edge costs are 1000*weight, snap weight 0.5


def test_libm_sincos_build_gives_the_same_search(small_cfg):
    """Swapping the fixed polynomial for libm's sin/cos (what the reference calls) must not change any
    discrete outcome on this fixture: the contract is about reproducibility, not about different answers."""
    res = []
    for libm in (False, True):
        x = Oracle(small_cfg, libm=libm)
        x.set_goal_joint(small_cfg.goal, small_cfg.goal_tol)
        x.set_start(small_cfg.start)
        x.search_params(5.0, 1.0, 1.0, True, True, 3000, 2000)
        res.append(x.plan())
    assert res[0]["cost"] == res[1]["cost"]
    assert np.array_equal(res[0]["expansion_log"], res[1]["expansion_log"])


# --- path post-processing ---

def _free_line(o, cfg, n, step):
    P = np.tile(np.array(cfg.start, float), (n, 1))
    P[:, 0] += step * np.arange(n)
    assert all(o.state_valid(q)[0] for q in P)
    return P


def test_shortcut_collapses_a_free_straight_line_and_keeps_endpoints(o, small_cfg):
    # shortcut.hpp:110-286: every extension (start, end+1) is valid and costs exactly the accumulated cost
    # (dyadic steps: no rounding), "<=" accepts it, so one direct edge remains
    P = _free_line(o, small_cfg, 9, 1.0 / 64)
    out, edge_checks, _ = o.post_process(P, shortcut=True, interpolate=False)
    assert np.array_equal(out, P[[0, -1]])
    assert edge_checks == len(P) - 1          # (0,1) then (0,2) ... (0,8)
    for n in (0, 1):
        out, _, _ = o.post_process(P[:n], shortcut=True, interpolate=False)
        assert np.array_equal(out, P[:n])
    out, _, _ = o.post_process(P[:2], shortcut=True, interpolate=False)
    assert np.array_equal(out, P[:2])


def test_shortcut_of_a_planned_path_is_a_valid_subsequence(small_cfg):
    x = Oracle(small_cfg)
    x.set_goal_joint(small_cfg.goal, small_cfg.goal_tol)
    x.set_start(small_cfg.start)
    x.search_params(5.0, 1.0, 1.0, True, True, 4000, 3000)
    r = x.plan()
    assert r["ok"]
    P = np.array([x.get_state(int(i))[0] if i != 0 else np.array(small_cfg.goal) for i in r["path"]])
    out, _, _ = x.post_process(P, shortcut=True, interpolate=False)
    assert 2 <= len(out) < len(P)
    assert np.array_equal(out[0], P[0]) and np.array_equal(out[-1], P[-1])
    k = 0
    for q in out:                              # points of the input, in order
        while not np.array_equal(P[k], q):
            k += 1
    for a, b in zip(out[:-1], out[1:]):
        assert x.edge_valid(a, b)[0]


def test_interpolate_path_fork_limit_test_and_upstream_behaviour(o, small_cfg):
    P = _free_line(o, small_cfg, 3, 0.25)
    # [FORK] collision_space.cpp:592-597: an in-limits segment reports "Joint limits violated": the path stays as it was
    out, _, checks = o.post_process(P, shortcut=False, interpolate=True)
    assert np.array_equal(out, P) and checks == 0
    # upstream test: every segment is replaced by its waypoints (post_processing.cpp:464-523), ends kept
    out, _, checks = o.post_process(P, shortcut=False, interpolate=True, upstream_limits=True)
    W = [o.waypoint_count(P[i], P[i + 1]) for i in range(2)]
    assert len(out) == 1 + sum(w - 1 for w in W) and checks == sum(W)
    assert np.array_equal(out[0], P[0]) and np.array_equal(out[W[0] - 1], P[1]) and np.array_equal(out[-1], P[2])
    assert np.all(np.diff(out[:, 0]) > 0)


def test_reference_cell_layout_variant_gives_the_same_results(small_cfg):
    """The oracle's timing variant with the reference's 48-byte array-of-structures cells over the padded grid
    (distance_map.h:110-127) must be the same checker: flags, lookups and a bounded search identical."""
    from oracle_binding import Oracle
    from smpl_amd import scenes
    a, b = Oracle(small_cfg), Oracle(small_cfg)
    b.use_aos_cells(True)
    for o in (a, b):
        o.set_goal_joint(small_cfg.goal, small_cfg.goal_tol)
    for q in scenes.random_states(scenes.ARM7_LIMITS, 40, 21):
        x, y = a.eval_state(q), b.eval_state(q)
        assert np.array_equal(x["flags"], y["flags"]) and np.array_equal(x["lookups"], y["lookups"]) and np.array_equal(x["h"], y["h"])
    for o in (a, b):
        o.set_start(small_cfg.start)
        o.search_params(5.0, 1.0, 1.0, True, True, 800, 800)
    ra, rb = a.plan(), b.plan()
    assert ra["cost"] == rb["cost"] and np.array_equal(ra["expansion_log"], rb["expansion_log"])
