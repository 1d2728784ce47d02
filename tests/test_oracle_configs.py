"""BASELINE config 1 ("smpl_test 7-DOF arm, ManipLattice + BFS heuristic, 128^3 voxel grid, single start/goal on CPU
(plumbing, no GPU)") run on the oracle: the whole CPU path -- scene, BFS, ARA* eps 100 -> 1 -- end to end, with
the invariants a plan must satisfy.  The same query runs on the GPU in tests/test_gpu_configs.py."""
import numpy as np
import pytest

from smpl_amd import scenes


@pytest.fixture(scope="module")
def cfg1():
    return scenes.config1()


def test_config1_scene_matches_the_survey(cfg1):
    g = cfg1.grid
    assert g.dims == (128, 128, 128) and g.res == 0.02 and g.origin == (-0.75, -1.28, 0.0) and g.max_dist == 0.4
    assert cfg1.params.eps0 == 100.0 and cfg1.params.cost_per_cell == 250 and cfg1.params.bfs_radius == 0.02
    assert cfg1.params.xy_rotate_by_var3 is True            # the fork's applyMotionPrimitive is the default
    assert cfg1.start == [0.0, 0.0, 0.0, -1.1356, 0.0, -1.05, 0.0]     # smpl_test/experiments/pr2_goal.yaml
    # the tabletop of smpl_test/env/tabletop.env is in the field: cells inside the slab are occupied
    c = scenes.world_to_grid(g.origin, g.res, (0.55, 0.0, 0.6))
    assert g.d2[tuple(c)] == 0
    assert g.d2.max() == g.dmax_int ** 2


def test_config1_plans_on_the_oracle(cfg1):
    from oracle_binding import Oracle
    o = Oracle(cfg1)
    assert o.M == 25      # 3 adaptive slots + (4 long + 7 short rows) x 2 (manip_lattice_action_space.cpp:218-261)
    o.set_goal_joint(cfg1.goal, cfg1.goal_tol)
    assert o.set_start(cfg1.start) == 1                     # id 0 is the goal (manip_lattice.cpp:122)
    p = cfg1.params
    o.search_params(p.eps0, p.eps_final, p.eps_delta, True, True, 4000, 4000)
    r = o.plan()
    assert r["ok"] == 1 and r["expansions"] <= 4000
    path = r["path"]
    assert path[0] == 1 and path[-1] == 0
    # cost = sum of the edge costs along the path: 1000 per file primitive (weight 1.0), 500 for the adaptive snap
    # (weight 0.5, manip_lattice_action_space.cpp:238-256; manip_lattice.cpp:1414-1437)
    total = 0
    for a, b in zip(path[:-1], path[1:]):
        succs, costs = o.get_succs(int(a))
        assert int(b) in succs
        total += int(min(c for s_, c in zip(succs, costs) if s_ == b))
    assert r["cost"] == total
    # the states on the path are collision-free and so are the edges between them
    Q = [o.get_state(int(i))[0] for i in path[:-1]]
    for a, b in zip(Q[:-1], Q[1:]):
        assert o.state_valid(b)[0]
        assert o.edge_valid(a, b)[0]
    # the log holds every expansion once per iteration, first the start
    assert r["expansion_log"][0] == 1 and len(r["expansion_log"]) == r["expansions"]
    # a second identical query on a fresh context gives the identical search (determinism of the whole CPU path)
    o2 = Oracle(cfg1)
    o2.set_goal_joint(cfg1.goal, cfg1.goal_tol)
    o2.set_start(cfg1.start)
    o2.search_params(p.eps0, p.eps_final, p.eps_delta, True, True, 4000, 4000)
    r2 = o2.plan()
    assert r2["cost"] == r["cost"] and np.array_equal(r2["expansion_log"], r["expansion_log"])


def test_config4_query_list_is_seeded_and_sharded():
    s1, g1 = scenes.config4_candidates()
    s2, g2 = scenes.config4_candidates()
    assert np.array_equal(s1, s2) and np.array_equal(g1, g2) and s1.shape[0] >= 2048
    ok = np.ones(s1.shape[0], bool)
    ok[::7] = False
    S, G = scenes.config4_queries(s1, g1, ok, ok)
    assert S.shape == (1024, 7) and not np.array_equal(S[0], s1[0])       # the rejected first candidate is skipped
    # whole cells from the cfg-2 start / goal
    cells = (S - np.asarray(scenes.ARM7_START)) / scenes.DEG
    assert np.allclose(cells, np.round(cells), atol=1e-9)
    assert [scenes.shard_range(r, 8) for r in (0, 1, 7)] == [(0, 128), (128, 256), (896, 1024)]
    with pytest.raises(ValueError):
        scenes.config4_queries(s1, g1, ok & False, ok)


def test_mt19937_64_known_answer_and_benchmark_states():
    """K2 micro-benchmark inputs follow the reference's scheme (benchmark_cc.cpp:280-301) with std::mt19937_64;
    the generator is pinned by the C++ standard's known answer (10000th output of seed 5489)."""
    g = scenes.MT19937_64()
    assert int(g.raw(10000)[-1]) == 9981545732273789042
    # split draws give the same stream as one draw
    a = scenes.MT19937_64(12345).raw(1000)
    h = scenes.MT19937_64(12345)
    b = np.concatenate([h.raw(7), h.raw(312), h.raw(681)])
    assert np.array_equal(a, b)
    Q = scenes.benchmark_states(scenes.ARM7_LIMITS, 4096)
    lo = np.array([l for l, _ in scenes.ARM7_LIMITS]); hi = np.array([h_ for _, h_ in scenes.ARM7_LIMITS])
    assert Q.shape == (4096, 7) and np.all(Q >= lo) and np.all(Q < hi)
    assert np.array_equal(Q, scenes.benchmark_states(scenes.ARM7_LIMITS, 4096))
    assert abs(Q[:, 4].mean()) < 0.15 and Q[:, 4].std() > 1.6          # uniform on [-pi, pi]
