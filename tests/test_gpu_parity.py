"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs.

Bars: bit-exact for every integer/boolean/index result (validity bits, waypoint counts, lookup counts,
coordinates, heuristic values, state ids, expansion order, path cost) and -- because of the arithmetic
contract -- bit-exact for the double results too (sphere positions, successor
joint values); north_star only asks 1e-6 for floats, asserted as well.
"""
import numpy as np
import pytest

from smpl_amd import scenes

pytestmark = pytest.mark.gpu

DEG = scenes.DEG


@pytest.fixture(scope="module", params=["pipeline", "fused", "small", "generic"])
def ctx(small_cfg, request):
    from oracle_binding import Oracle
    from smpl_amd import capi
    if capi.lib().smplx_device_count() == 0:
        pytest.fail("no GPU visible: the gpu-marked tests must run on the MI355X box")
    o = Oracle(small_cfg)
    o.set_order(chain=True)   # the kernel walks the sphere trees link by link (same booleans, see oracle)
    # "pipeline": four-kernel waypoint-parallel path forced for every batch size; "small": batches <= 256 states take
    # the single-launch kernel (the default policy); "fused": one thread per edge
    # "generic": the kernels linked into the library; the other three run the per-robot hiprtc build of the same source
    s = capi.Space.from_config(small_cfg, fused=(request.param == "fused"),
                               no_small_kernel=(request.param in ("pipeline", "generic")),
                               generic_kernels=(request.param == "generic"))
    ok, note = s.specialized()
    if request.param == "generic":
        assert not ok
    # (that the per-robot build is actually in use is asserted once, in tests/test_zz_per_robot_build.py)
    s.fused = request.param == "fused"
    o.set_goal_joint(small_cfg.goal, small_cfg.goal_tol)
    s.set_goal_joint(small_cfg.goal, small_cfg.goal_tol)
    return small_cfg, o, s


def _random_states(n, seed):
    return scenes.random_states(scenes.ARM7_LIMITS, n, seed)


def test_goal_pose_and_bfs_grid(ctx):
    cfg, o, s = ctx
    assert np.array_equal(o.goal_pose(), s.goal_pose())
    assert np.array_equal(o.bfs_grid(), s.bfs_grid())


def test_bfs_grid_over_ten_goals_in_one_space(small_cfg):
    """The distances carry the tag of the run that wrote them (device_types.h SmplxBfsDev) and the records are reset only
    when the tags wrap, at the eighth goal: every goal's grid must equal the oracle's all the same, including one whose
    cell is out of the grid (bfs3d.cpp:169-171: nothing is labelled) right after a labelled one."""
    from oracle_binding import Oracle
    from smpl_amd import capi
    o = Oracle(small_cfg)
    s = capi.Space.from_config(small_cfg)
    rng = np.random.default_rng(17)
    for k in range(10):
        g = np.array(small_cfg.goal) + rng.uniform(-0.4, 0.4, size=len(small_cfg.goal))
        o.set_goal_joint(list(g), small_cfg.goal_tol)
        s.set_goal_joint(list(g), small_cfg.goal_tol)
        exp = o.bfs_grid()
        assert np.array_equal(exp, s.bfs_grid()), f"goal {k}"
        assert ((exp >= 0) & (exp < 0x7FFFFFFF)).sum() > 1000, "the goal cell is free and the BFS spreads"
        if k == 4:
            far = [50.0, 50.0, 50.0]
            o.set_goal_xyz(far, small_cfg.goal_tol)
            s.set_goal_xyz(far, small_cfg.goal_tol)
            out = s.bfs_grid()
            assert np.array_equal(o.bfs_grid(), out), "out-of-grid goal"
            assert not ((out >= 0) & (out < 0x7FFFFFFF)).any()


def test_state_validity_with_deeper_sphere_trees(small_cfg):
    """A robot whose trees need a 24-byte traversal stack (the default is 16): verdicts and lookup tallies of random states
    and edges still equal the oracle's."""
    import copy
    from oracle_binding import Oracle
    from smpl_amd import capi
    cfg = copy.copy(small_cfg)
    cfg.robot_text = scenes.with_extra_spheres(small_cfg.robot_text, 14)
    o = Oracle(cfg)
    o.set_order(chain=True)
    s = capi.Space.from_config(cfg)
    Q = _random_states(1500, 21)
    ok, lk = s.state_valid_batch(Q)
    exp = [o.state_valid(q) for q in Q]
    assert np.array_equal(ok.astype(bool), np.array([e[0] for e in exp]))
    assert np.array_equal(lk, np.array([e[1] for e in exp]))
    assert 0.02 < ok.mean() < 0.98
    o.set_goal_joint(cfg.goal, cfg.goal_tol)
    s.set_goal_joint(cfg.goal, cfg.goal_tol)
    _compare_expand(o, s, Q[ok.astype(bool)][:40])


def test_bfs_grid_from_goal_cells_at_brick_corners(small_cfg):
    """The BFS records keep, for every 8x8x8 brick, copies of the eight cells diagonally across its corners, written by the
    bricks that own them: goals ON such corner cells (the seed writes the copy) and next to them, whole grid against the
    oracle's."""
    from oracle_binding import Oracle
    from smpl_amd import capi
    o = Oracle(small_cfg)
    s = capi.Space.from_config(small_cfg)
    o.set_goal_joint(small_cfg.goal, small_cfg.goal_tol)
    walls = o.bfs_grid().reshape(-1)
    n = small_cfg.grid.dims
    org, res = np.array(small_cfg.grid.origin), small_cfg.grid.res
    dx, dy = n[0] + 2, n[1] + 2
    done = 0
    for base in ((8, 8, 8), (16, 24, 8), (24, 16, 16), (32, 32, 24), (40, 8, 32)):
        for d in ((-1, -1, -1), (0, 0, 0), (-1, 0, -1), (0, -1, 0)):
            c = tuple(b + e for b, e in zip(base, d))
            if not all(0 <= c[a] < n[a] for a in range(3)):
                continue
            if walls[(c[2] + 1) * dx * dy + (c[1] + 1) * dx + (c[0] + 1)] == 0x7FFFFFFF:
                continue       # a wall cell there in this scene
            xyz = list(org + res * (np.array(c) + 0.5))
            o.set_goal_xyz(xyz, small_cfg.goal_tol)
            s.set_goal_xyz(xyz, small_cfg.goal_tol)
            assert np.array_equal(o.bfs_grid(), s.bfs_grid()), f"goal cell {c}"
            done += 1
    assert done >= 8


def test_sphere_positions_bitwise(ctx):
    cfg, o, s = ctx
    Q = _random_states(64, 1)
    got = s.sphere_positions(Q)
    for i in range(Q.shape[0]):
        exp = o.sphere_positions(Q[i], s.model.nnodes)
        assert np.array_equal(exp, got[i]), f"state {i}"
        assert np.max(np.abs(exp - got[i])) <= 1e-6


def test_state_valid_batch(ctx):
    cfg, o, s = ctx
    Q = _random_states(2000, 2)
    ok, lk = s.state_valid_batch(Q)
    exp = [o.state_valid(q) for q in Q]
    assert np.array_equal(ok.astype(bool), np.array([e[0] for e in exp]))
    assert np.array_equal(lk, np.array([e[1] for e in exp]))
    assert 0.05 < ok.mean() < 0.95   # the sample exercises both outcomes


def test_edge_valid_batch_short_and_long(ctx):
    cfg, o, s = ctx
    rng = np.random.default_rng(3)
    A = _random_states(1500, 3)
    B = A.copy()
    # primitive-sized moves on one joint
    j = rng.integers(0, 7, size=A.shape[0])
    B[np.arange(A.shape[0]), j] += rng.choice([-7, -4, 4, 7], size=A.shape[0]) * DEG
    # long edges (stride-5 waypoint order), including wrap-around of the continuous joints
    B[1000:] = _random_states(500, 4)
    B[1400:, 4] += 2.5 * np.pi
    B[1450:] = A[1450:]          # zero motion: no waypoints
    ok, lk, w = s.edge_valid_batch(A, B)
    eo, el = o.edge_valid_batch(A, B)
    ew = np.array([o.waypoint_count(a, b) for a, b in zip(A, B)])
    assert np.array_equal(w, ew)
    assert np.array_equal(ok, eo)
    assert np.array_equal(lk, el)
    assert w.max() > 5 and (w == 0).sum() == 50


def test_interpolate_matches_oracle_waypoints(ctx):
    cfg, o, s = ctx
    a = np.array(cfg.start); b = a.copy(); b[0] += 7 * DEG
    pts, n = s.interpolate(a, b)
    assert n == o.waypoint_count(a, b)
    assert np.array_equal(pts[0], a)


def test_heuristic_batch(ctx):
    cfg, o, s = ctx
    Q = _random_states(1000, 5)
    h, xyz = s.heuristic_batch(Q)
    assert np.array_equal(h, np.array([o.heuristic_q(q) for q in Q]))
    assert np.array_equal(xyz, np.array([o.planning_fk(q) for q in Q]))
    assert len(np.unique(h)) > 10


def _compare_expand(o, s, Q):
    got = s.expand_batch(Q)
    for i, q in enumerate(Q):
        exp = o.eval_state(q)
        assert np.array_equal(exp["flags"], got["flags"][i]), f"flags of state {i}"
        valid = (exp["flags"] & 1) != 0
        evaluated = (exp["flags"] & 0x10) == 0
        assert np.array_equal(exp["coord"][valid], got["coord"][i][valid])
        assert np.array_equal(exp["q"][evaluated], got["q"][i][evaluated])
        assert np.array_equal(exp["h"][valid], got["h"][i][valid])
        assert np.array_equal(exp["cost"][valid], got["cost"][i][valid])
        if getattr(s, "fused", False):
            # one thread per edge, reference waypoint order and early exit: tallies identical everywhere
            assert np.array_equal(exp["lookups"], got["lookups"][i]), f"lookups of state {i}"
        else:
            # waypoint-parallel: identical wherever the edge is not in collision; a colliding edge has all its
            # waypoints examined, so its tally can only be larger
            coll = (exp["flags"] & 0x40) != 0
            assert np.array_equal(exp["lookups"][~coll], got["lookups"][i][~coll]), f"lookups of state {i}"
            assert np.all(got["lookups"][i][coll] >= exp["lookups"][coll])
    return got


def test_expand_batch_random_states(ctx):
    cfg, o, s = ctx
    Q = _random_states(300, 6)
    got = _compare_expand(o, s, Q)
    f = got["flags"]
    assert (f & 1).sum() > 100 and (f & 0x40).sum() > 100 and (f & 0x20).sum() > 10


def test_expand_batch_near_goal_and_start(ctx):
    cfg, o, s = ctx
    g = np.array(cfg.goal)
    Q = [np.array(cfg.start), g]
    for k in range(7):
        q = g.copy(); q[k] += 4 * DEG; Q.append(q)
        q = g.copy(); q[k] -= 4 * DEG; Q.append(q)
    got = _compare_expand(o, s, np.array(Q))
    assert (got["flags"] & 2).sum() >= 7   # goal successors (short primitive back, or the snap)


def test_expand_batch_upstream_semantics_switch(small_cfg):
    """xy_rotate_by_var3 = 0: upstream smpl's applyMotionPrimitive (no rotation), kept as an opt-out."""
    from oracle_binding import Oracle
    from smpl_amd import capi
    o = Oracle(small_cfg, xy_rotate=False)
    o.set_order(chain=True)
    s = capi.Space.from_config(small_cfg, xy_rotate=False)
    s.fused = False
    o.set_goal_joint(small_cfg.goal, small_cfg.goal_tol)
    s.set_goal_joint(small_cfg.goal, small_cfg.goal_tol)
    _compare_expand(o, s, _random_states(100, 8))


def test_getsuccs_ids_match_sequential_reference_order(ctx):
    cfg, o, s = ctx
    o.set_goal_joint(cfg.goal, cfg.goal_tol); s.set_goal_joint(cfg.goal, cfg.goal_tol)   # fresh query on both sides
    assert o.set_start(cfg.start) == s.set_start(cfg.start) == 1
    frontier = [1]
    seen = 0
    for _ in range(40):
        i = frontier[seen]; seen += 1
        es, ec = o.get_succs(i)
        gs, gc = s.get_succs(i)
        assert np.array_equal(es, gs) and np.array_equal(ec, gc)
        frontier += [int(x) for x in es if x != 0 and x not in frontier]
    assert o.num_states() == s.num_states()
    for i in range(1, s.num_states()):
        eq, ecd = o.get_state(i)
        gq, gcd = s.get_state(i)
        assert np.array_equal(eq, gq) and np.array_equal(ecd, gcd)
        assert o.heuristic_q(eq) == s.goal_heuristic(i)


@pytest.mark.parametrize("semantics", ["fork", "upstream"])
@pytest.mark.parametrize("fused", [False, True, "pipeline-only", "host-loop"])
@pytest.mark.parametrize("goal_kind", ["joint", "xyz"])
def test_arastar_expansion_order_and_cost(small_cfg, goal_kind, fused, semantics, monkeypatch):
    """fused False: the default path (device-resident search); True / pipeline-only / host-loop: the host-driven ARA* with
    the fused kernels, the four-kernel pipeline for every batch, the default kernels."""
    from oracle_binding import Oracle
    from smpl_amd import capi
    cfg = small_cfg
    if fused == "host-loop":
        monkeypatch.setenv("SMPLX_SEARCH", "host")
    rot = semantics == "fork"
    o = Oracle(cfg, xy_rotate=rot)
    s = capi.Space.from_config(cfg, batch_states=256, xy_rotate=rot, fused=(fused is True),
                               no_small_kernel=(fused == "pipeline-only"))
    if goal_kind == "joint":
        o.set_goal_joint(cfg.goal, cfg.goal_tol); s.set_goal_joint(cfg.goal, cfg.goal_tol)
    else:
        p = o.planning_fk(cfg.goal)
        o.set_goal_xyz(p, [0.04] * 3); s.set_goal_xyz(p, [0.04] * 3)
    assert o.set_start(cfg.start) == s.set_start(cfg.start)
    o.search_params(5.0, 1.0, 1.0, True, True, 6000, 3000)
    eo = o.plan()
    go = s.plan(5.0, 1.0, 1.0, True, True, 6000, 3000)
    assert eo["ok"] == go["solved"]
    assert eo["expansions"] == go["expansions"]
    assert np.array_equal(eo["expansion_log"], go["expansion_log"])   # expanded-state set AND order, ids bit-exact
    assert eo["cost"] == go["cost"]
    assert np.array_equal(eo["path"], go["path"])
    assert eo["eps"] == go["satisfied_eps"]
    assert o.num_states() == s.num_states()
    assert eo["succ_evals"] == go["committed_succ_evals"]
    if go["solved"]:
        q = s.extract_path(go["path"])
        assert np.array_equal(q[0], np.array(cfg.start))


@pytest.fixture(scope="module")
def planned_path(small_cfg):
    from smpl_amd import capi
    cfg = small_cfg
    s = capi.Space.from_config(cfg, batch_states=256)
    s.set_goal_joint(cfg.goal, cfg.goal_tol)
    s.set_start(cfg.start)
    go = s.plan(5.0, 1.0, 1.0, True, True, 6000, 3000)
    assert go["solved"]
    return s.extract_path(go["path"])


@pytest.mark.parametrize("mode", [(True, True, False), (True, True, True), (True, False, True), (False, True, True)])
def test_post_process_path_equals_reference_loops(ctx, planned_path, mode):
    """Row N3: shortcut + interpolation of a planned path and of a random zig-zag of valid states: the point
    sequences are the oracle's, bit for bit."""
    cfg, o, s = ctx
    paths = [planned_path]
    Q = _random_states(400, 77)
    ok, _ = s.state_valid_batch(Q)
    paths.append(Q[ok.astype(bool)][:30])
    paths += [paths[0][:1], paths[0][:2], paths[0][:0]]
    for P in paths:
        want, _, _ = o.post_process(P, *mode)
        got, st = s.post_process_path(P, *mode)
        assert got.shape == want.shape
        assert np.array_equal(got, want)
    # the fork's limit test leaves in-limits paths un-interpolated (collision_space.cpp:592-597)
    if mode == (False, True, False):
        assert np.array_equal(got, paths[-1])


@pytest.fixture(scope="module", params=["per-robot", "generic", "per-robot-pipeline", "fused"])
def mixed_ctx(request):
    """The robot that takes every branch of the joint-transform functions (rotated origins, generic axis, prismatic,
    continuous): its per-robot build keeps the joint records in LDS for the kinds without a literal form."""
    from oracle_binding import Oracle
    from smpl_amd import capi
    cfg = scenes.config_mixed()
    o = Oracle(cfg)
    o.set_order(chain=True)
    s = capi.Space.from_config(cfg, generic_kernels=(request.param == "generic"), fused=(request.param == "fused"),
                               no_small_kernel=(request.param == "per-robot-pipeline"))
    s.fused = request.param == "fused"
    o.set_goal_joint(cfg.goal, cfg.goal_tol)
    s.set_goal_joint(cfg.goal, cfg.goal_tol)
    return cfg, o, s


def test_mixed_joint_kinds_positions_validity_and_expansion(mixed_ctx):
    cfg, o, s = mixed_ctx
    Q = np.vstack([np.array(cfg.start), np.array(cfg.goal), scenes.random_states(scenes.MIXED_LIMITS, 120, 9)])
    nn = o.model()["xyzr"].shape[0]
    pos = s.sphere_positions(Q)
    for i, q in enumerate(Q[:40]):
        assert np.array_equal(pos[i], o.sphere_positions(q, nn))      # every transform kind, bit for bit
    ok, lk = s.state_valid_batch(Q)
    for i, q in enumerate(Q):
        eo, el = o.state_valid(q)
        assert bool(ok[i]) == eo
        if eo:
            assert lk[i] == el
    V = Q[ok.astype(bool)][:48]
    _compare_expand(o, s, V)


def test_mixed_joint_kinds_search(mixed_ctx):
    cfg, o, s = mixed_ctx
    assert o.set_start(cfg.start) == s.set_start(cfg.start)
    o.search_params(5.0, 1.0, 1.0, True, True, 3000, 3000)
    eo = o.plan()
    go = s.plan(5.0, 1.0, 1.0, True, True, 3000, 3000)
    assert eo["ok"] == go["solved"] and eo["cost"] == go["cost"]
    assert np.array_equal(eo["expansion_log"], go["expansion_log"])
    assert np.array_equal(eo["path"], go["path"])


@pytest.fixture(scope="module")
def dual_ctx():
    """14-DOF dual arm (SURVEY cfg 5 robot) on a coarse grid: two kinematic chains from the root, 16 sphere trees,
    85 checked link pairs including every inter-arm pair -- exercises the sphere-sphere slow path."""
    from oracle_binding import Oracle
    from smpl_amd import capi
    cfg = scenes.config5(n=64, nboxes=12, res=0.08)
    o = Oracle(cfg)
    o.set_order(chain=True)
    s = capi.Space.from_config(cfg)
    s.fused = False
    o.set_goal_joint(cfg.goal, cfg.goal_tol)
    s.set_goal_joint(cfg.goal, cfg.goal_tol)
    return cfg, o, s


def _dual_states(n, seed):
    lim = scenes.ARM7_LIMITS + scenes.ARM7_LIMITS
    return scenes.random_states(lim, n, seed)


def test_dual_arm_state_validity_and_self_collision(dual_ctx):
    cfg, o, s = dual_ctx
    assert (s.model.nvars, s.model.ntrees, s.model.npairs) == (14, 16, 85)
    Q = _dual_states(1500, 31)
    # fold the left arm towards the right one so inter-arm pairs actually collide
    Q[:500, 7] = -Q[:500, 0]
    ok, lk = s.state_valid_batch(Q)
    exp = [o.state_valid(q) for q in Q]
    assert np.array_equal(ok.astype(bool), np.array([e[0] for e in exp]))
    assert np.array_equal(lk, np.array([e[1] for e in exp]))
    assert 0.02 < ok.mean() < 0.98
    assert np.array_equal(o.bfs_grid(), s.bfs_grid())


def test_dual_arm_expand_batch(dual_ctx):
    cfg, o, s = dual_ctx
    Q = np.vstack([np.array(cfg.start), _dual_states(60, 32)])
    got = _compare_expand(o, s, Q)
    assert got["flags"].shape[1] == 59          # 3 adaptive slots + 28 rows x 2
    assert (got["flags"] & 1).sum() > 50


@pytest.mark.parametrize("shared_scene", ["device", True, 3, 1, False])
def test_interleaved_multi_query_equals_each_query_alone(small_cfg, shared_scene, monkeypatch):
    """smplx_plan_multi: independent queries side by side on one GPU (BASELINE config 4 shape).  Every query must come
    out exactly as it does alone -- and as the oracle computes it.  "device": the default, one persistent workgroup per
    query (the searches themselves are in tests/test_gpu_device_search.py); the other four the host-driven loop
    (SMPLX_SEARCH=host) with its thread layouts."""
    from oracle_binding import Oracle
    from smpl_amd import capi
    cfg = small_cfg
    on_device = shared_scene == "device"
    if not on_device:
        monkeypatch.setenv("SMPLX_SEARCH", "host")
    cells = [[-49, 7, 21, -14, -8, -12, 16], [-21, 7, 14, -7, 8, -4, 12], [-35, 14, 7, -14, 4, -8, 8], [-42, 10, 14, -10, 0, -8, 12]]
    goals = [[cfg.start[i] + c * DEG for i, c in enumerate(cs)] for cs in cells]
    spaces = []
    # shared grid + model handles -> the misses of all queries ride in ONE cross-query launch per sweep;
    # separate handles -> every query issues its own batches on its own stream
    grid = capi.Grid(cfg.grid.origin, cfg.grid.dims, cfg.grid.res, cfg.grid.max_dist, cfg.grid.d2)
    model = capi.Model(cfg.robot_text)
    for g in goals:
        sp = (capi.Space(model, grid, cfg.mprim, cfg.params, 512) if shared_scene
              else capi.Space.from_config(cfg, batch_states=512))
        sp.set_goal_joint(g, cfg.goal_tol)
        sp.set_start(cfg.start)
        spaces.append(sp)
    # shared scene: True -> 2 host threads, 3 -> 3 (asynchronous driver: worker threads + one GPU submitter), 1 -> one
    # thread (sweeps of one cross-query batch); separate scenes: every query its own batches
    threads = 1 if on_device else (2 if shared_scene is True else (shared_scene if shared_scene else 1))
    multi, wall = capi.Space.plan_multi(spaces, 5.0, 1.0, 1.0, True, True, 4000, 2500, host_threads=threads)
    assert wall > 0 and len(multi) == 4
    for g, m in zip(goals, multi):
        o = Oracle(cfg)
        o.set_goal_joint(g, cfg.goal_tol)
        o.set_start(cfg.start)
        o.search_params(5.0, 1.0, 1.0, True, True, 4000, 2500)
        e = o.plan()
        assert e["ok"] == m["solved"] and e["cost"] == m["cost"] and e["expansions"] == m["expansions"]
        assert np.array_equal(e["expansion_log"], m["expansion_log"])
        assert np.array_equal(e["path"], m["path"])
    if on_device:
        assert spaces[0].search_counters()["searches"] > 0
        return
    assert sum(m["gpu_batches"] for m in multi) > 4
    if shared_scene:
        # host_threads = 2: worker threads own the searches, ONE submitter thread issues the cross-query batches
        # (accounted to the leading space, query 0)
        assert multi[0]["gpu_batches"] > 0
        assert all(m["gpu_batches"] == 0 for m in multi[1:])


def test_deferred_pass_when_the_work_list_overflows(small_cfg):
    """Edges whose waypoints do not fit the work list are resolved by the fused pass inside the same call
    (SMPLX_F_DEFERRED never leaks out); results identical to the oracle."""
    from oracle_binding import Oracle
    from smpl_amd import capi
    o = Oracle(small_cfg)
    o.set_order(chain=True)
    s = capi.Space.from_config(small_cfg, tiny_work_list=True)
    s.fused = False
    o.set_goal_joint(small_cfg.goal, small_cfg.goal_tol)
    s.set_goal_joint(small_cfg.goal, small_cfg.goal_tol)
    Q = np.vstack([np.array(small_cfg.start), _random_states(200, 41)])
    got = s.expand_batch(Q)
    assert not (got["flags"] & 0x80).any()
    for i, q in enumerate(Q):
        exp = o.eval_state(q)
        assert np.array_equal(exp["flags"], got["flags"][i])
        v = (exp["flags"] & 1) != 0
        assert np.array_equal(exp["coord"][v], got["coord"][i][v]) and np.array_equal(exp["h"][v], got["h"][i][v])


def test_distance_cap_below_the_sphere_radii(small_cfg):
    """A distance field capped at 0.12 m (3 cells) against spheres of up to 0.15 m: the cap makes a big sphere
    "collide" wherever it is (distance_map.hpp:281-300 caps at dmax; collision_operations.h:67-77), so the traversal
    always descends and a big LEAF always fails.  Thresholds beyond the cap take the dmax^2 + 1 sentinel."""
    import copy
    from oracle_binding import Oracle
    from smpl_amd import capi
    cfg = copy.copy(small_cfg)
    g = small_cfg.grid
    cfg.grid = scenes.build_grid(g.origin, g.dims, g.res, 0.12, small_cfg.boxes)
    o = Oracle(cfg)
    o.set_order(chain=True)
    s = capi.Space.from_config(cfg, fused=True)
    Q = _random_states(100, 59)
    ok, lk = s.state_valid_batch(Q)
    for i, q in enumerate(Q):
        eo, el = o.state_valid(q)
        assert bool(ok[i]) == eo and lk[i] == el     # one thread per state: the reference's early exit, tally for tally
    assert not ok.any()                              # the 0.15 m shoulder leaf can never clear a 0.12 m cap


@pytest.mark.parametrize("padding", [0.013, 0.05])
def test_sphere_padding(small_cfg, padding):
    """SelfCollisionModel's m_padding (collision_operations.h:67-77: valid iff dist^2 >= (r + pad)^2): the engine folds
    it into the integer thresholds (model_compile.cpp sphere_threshold)."""
    from oracle_binding import Oracle
    from smpl_amd import capi
    o = Oracle(small_cfg)
    o.set_order(chain=True)
    o.set_padding(padding)
    s = capi.Space.from_config(small_cfg, padding=padding)
    s.fused = False
    o.set_goal_joint(small_cfg.goal, small_cfg.goal_tol)
    s.set_goal_joint(small_cfg.goal, small_cfg.goal_tol)
    Q = _random_states(300, 53)
    ok, lk = s.state_valid_batch(Q)
    exp = [o.state_valid(q) for q in Q]
    assert np.array_equal(ok.astype(bool), np.array([e[0] for e in exp]))
    ok0, _ = capi.Space.from_config(small_cfg).state_valid_batch(Q)
    assert ok.sum() < ok0.sum()                     # the padding does reject states the bare spheres accept
    _compare_expand(o, s, Q[ok.astype(bool)][:40])


def test_reference_mprim_file_through_the_c_abi(small_cfg):
    """The reference's own smpl_test/config/pr2.mprim (tests/golden/pr2.mprim, upstream row format) handed to
    smplx_space_create as it is."""
    import copy
    import os
    from oracle_binding import Oracle
    from smpl_amd import capi
    cfg = copy.copy(small_cfg)
    cfg.mprim = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pr2.mprim")).read()
    o = Oracle(cfg)
    o.set_order(chain=True)
    s = capi.Space.from_config(cfg)
    s.fused = False
    assert s.M == o.M == 25
    o.set_goal_joint(cfg.goal, cfg.goal_tol)
    s.set_goal_joint(cfg.goal, cfg.goal_tol)
    _compare_expand(o, s, np.vstack([np.array(cfg.start), _random_states(40, 61)]))


@pytest.mark.parametrize("long_and_short", [False, True])
def test_fork_mprim_rows_with_weights_and_long_and_short_gating(small_cfg, long_and_short):
    """[FORK] .mprim rows carry a group and a weight after the deltas (manip_lattice_action_space.cpp:149,161-186):
    the weight scales the edge cost (manip_lattice.cpp:1414-1437); use_long_and_short keeps both families active
    (:662-691)."""
    import copy
    from oracle_binding import Oracle
    from smpl_amd import capi
    cfg = copy.copy(small_cfg)
    rows = []
    for j, (cells, w) in enumerate([(7, 1.0), (7, 2.5), (6, 0.5), (5, 1.25)]):
        r = [0] * 7; r[j] = cells
        rows.append(" ".join(map(str, r)) + f" {j % 2} {w}")
    for j, w in enumerate([1.0, 0.75, 3.0, 1.0, 0.2, 1.0, 1.5]):
        r = [0] * 7; r[j] = 3
        rows.append(" ".join(map(str, r)) + f" 1 {w}")
    cfg.mprim = "Motion_Primitives(degrees): 11 9 7\n" + "\n".join(rows) + "\n"
    cfg.params = copy.copy(cfg.params)
    cfg.params.use_long_and_short = long_and_short
    o = Oracle(cfg)
    o.set_order(chain=True)
    s = capi.Space.from_config(cfg)
    s.fused = False
    o.set_goal_joint(cfg.goal, cfg.goal_tol)
    s.set_goal_joint(cfg.goal, cfg.goal_tol)
    near = np.array(cfg.goal) + np.array([2, -1, 1, 0, 3, -2, 1]) * DEG
    Q = np.vstack([np.array(cfg.start), near, _random_states(60, 47)])
    got = _compare_expand(o, s, Q)
    costs = set(np.unique(got["cost"][(got["flags"] & 1) != 0]).tolist())
    assert {2500, 500, 1250, 750, 3000, 200} & costs           # weighted costs show up: int(1000 * weight)
    assert o.set_start(cfg.start) == s.set_start(cfg.start)
    o.search_params(5.0, 1.0, 1.0, True, True, 2500, 2500)
    eo = o.plan()
    go = s.plan(5.0, 1.0, 1.0, True, True, 2500, 2500)
    assert eo["cost"] == go["cost"] and np.array_equal(eo["expansion_log"], go["expansion_log"])


@pytest.mark.parametrize("mode", ["small", "pipeline", "generic"])
def test_long_edges_of_25_and_more_waypoints(small_cfg, mode):
    """Primitives of 25 and 40 cells: edges of up to ~35 waypoints.  The small-batch kernel wraps them around the 7
    waypoint lanes of an edge, the pipeline spreads them over the work list."""
    import copy
    from oracle_binding import Oracle
    from smpl_amd import capi
    cfg = copy.copy(small_cfg)
    cfg.mprim = scenes.mprim_text(7, [0, 1, 2, 3], [0, 1, 2, 3, 4, 5, 6], long_cells=40, short_cells=25)
    o = Oracle(cfg)
    o.set_order(chain=True)
    s = capi.Space.from_config(cfg, no_small_kernel=(mode != "small"), generic_kernels=(mode == "generic"))
    s.fused = False
    o.set_goal_joint(cfg.goal, cfg.goal_tol)
    s.set_goal_joint(cfg.goal, cfg.goal_tol)
    Q = np.vstack([np.array(cfg.start), _random_states(60, 43)])
    got = _compare_expand(o, s, Q)
    assert not (got["flags"] & 0x80).any()
    A = np.repeat(Q[:8], 4, axis=0)
    B = A.copy(); B[:, 0] += 40 * DEG
    _, _, w = s.edge_valid_batch(A, B)
    assert w.max() > 25


def test_edge_cases_empty_batches_bad_start_and_goal_outside_grid(small_cfg):
    from oracle_binding import Oracle
    from smpl_amd import capi
    cfg = small_cfg
    s = capi.Space.from_config(cfg)
    # empty batches are fine and touch nothing
    ok, lk = s.state_valid_batch(np.zeros((0, 7)))
    assert ok.shape == (0,)
    e, _, _ = s.edge_valid_batch(np.zeros((0, 7)), np.zeros((0, 7)))
    assert e.shape == (0,)
    # expansion needs a goal (the primitives are gated by goal distance): call-sequence error, not a crash
    with pytest.raises(capi.SmplxError) as err:
        s.expand_batch(np.array([cfg.start]))
    assert err.value.code == -5
    s.set_goal_joint(cfg.goal, cfg.goal_tol)
    assert s.expand_batch(np.zeros((0, 7)))["flags"].shape == (0, 25)
    # a start state outside the joint limits / in collision is refused like ManipLattice::setStart (:1956-1967)
    bad = np.array(cfg.start); bad[3] = 0.5
    with pytest.raises(capi.SmplxError) as err:
        s.set_start(bad)
    assert err.value.code == -6
    coll = scenes.random_states(scenes.ARM7_LIMITS, 400, 51)
    okc, _ = s.state_valid_batch(coll)
    with pytest.raises(capi.SmplxError) as err:
        s.set_start(coll[np.argmin(okc)])
    assert err.value.code == -6
    # goal position outside the grid: BFS_3D::run labels nothing (bfs3d.cpp:169-171), every heuristic is
    # cost_per_cell * UNDISCOVERED(-1) or Infinity for walls -- same as the oracle
    o = Oracle(cfg)
    far = [cfg.grid.origin[0] - 1.0, 0.0, 1.0]
    o.set_goal_xyz(far, [0.04] * 3); s.set_goal_xyz(far, [0.04] * 3)
    assert np.array_equal(o.bfs_grid(), s.bfs_grid())
    Q = _random_states(50, 52)
    h, _ = s.heuristic_batch(Q)
    assert np.array_equal(h, np.array([o.heuristic_q(q) for q in Q]))
    assert set(np.unique(h)) <= {-cfg.params.cost_per_cell, 32767}


def test_config3_pr2_like_cluttered_tabletop():
    """SURVEY cfg 3: 150^3 grid (not a multiple of the 4-cell brick), max_dist 1.8 m (squared cell distances up to
    8100 in 16 bits), clutter on the table, eps 100."""
    from oracle_binding import Oracle
    from smpl_amd import capi
    cfg = scenes.config3()
    assert cfg.grid.d2.max() > 400
    o = Oracle(cfg)
    o.set_order(chain=True)
    s = capi.Space.from_config(cfg, batch_states=1024)
    s.fused = False
    o.set_goal_joint(cfg.goal, cfg.goal_tol)
    s.set_goal_joint(cfg.goal, cfg.goal_tol)
    assert np.array_equal(o.bfs_grid(), s.bfs_grid())
    _compare_expand(o, s, np.vstack([np.array(cfg.start), _random_states(80, 61)]))
    o.set_order(chain=False)
    assert o.set_start(cfg.start) == s.set_start(cfg.start)
    o.search_params(100.0, 1.0, 1.0, True, True, 2500, 2500)
    eo = o.plan()
    go = s.plan(100.0, 1.0, 1.0, True, True, 2500, 2500)
    assert eo["ok"] == go["solved"] and eo["cost"] == go["cost"]
    assert np.array_equal(eo["expansion_log"], go["expansion_log"])
