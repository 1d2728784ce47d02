"""GPU parity at the BASELINE configs' own sizes: cfg 1 (128^3), cfg 2 (256^3, eps 5, B = 4096),
cfg 4 (the 128-query shard of one GPU), cfg 5 (14-DOF, 512^3).  cfg 3 is in test_gpu_parity.py.  Everything
through the C-ABI, against the oracle on the same seeded inputs; fork semantics (the default)."""
import numpy as np
import pytest

from smpl_amd import scenes

pytestmark = pytest.mark.gpu
DEG = scenes.DEG


def _need_gpu():
    from smpl_amd import capi
    if capi.lib().smplx_device_count() == 0:
        pytest.fail("no GPU visible: the gpu-marked tests must run on the MI355X box")


def _compare_batch(o, s, Q):
    got = s.expand_batch(Q)
    for i, q in enumerate(Q):
        e = o.eval_state(q)
        assert np.array_equal(e["flags"], got["flags"][i]), f"flags of state {i}"
        v = (e["flags"] & 1) != 0
        ev = (e["flags"] & 0x10) == 0
        assert np.array_equal(e["coord"][v], got["coord"][i][v])
        assert np.array_equal(e["q"][ev], got["q"][i][ev])
        assert np.array_equal(e["h"][v], got["h"][i][v]) and np.array_equal(e["cost"][v], got["cost"][i][v])
        coll = (e["flags"] & 0x40) != 0
        assert np.array_equal(e["lookups"][~coll], got["lookups"][i][~coll])
    return got


def _same_search(o, s, eps0, nb_init, nb_rep):
    o.search_params(eps0, 1.0, 1.0, True, True, nb_init, nb_rep)
    eo = o.plan()
    go = s.plan(eps0, 1.0, 1.0, True, True, nb_init, nb_rep)
    assert eo["ok"] == go["solved"] and eo["expansions"] == go["expansions"]
    assert np.array_equal(eo["expansion_log"], go["expansion_log"])       # expanded-state set and order, ids bit-exact
    assert eo["cost"] == go["cost"] and np.array_equal(eo["path"], go["path"])
    assert eo["eps"] == go["satisfied_eps"]
    assert o.num_states() == s.num_states()
    assert eo["succ_evals"] == go["committed_succ_evals"]
    return eo, go


def test_config1_128cube_on_the_gpu():
    """cfg 1's query (the CPU plumbing case) through the engine: same search as the oracle's."""
    from oracle_binding import Oracle
    from smpl_amd import capi
    _need_gpu()
    cfg = scenes.config1()
    o = Oracle(cfg)
    s = capi.Space.from_config(cfg, batch_states=1024)
    o.set_goal_joint(cfg.goal, cfg.goal_tol); s.set_goal_joint(cfg.goal, cfg.goal_tol)
    assert np.array_equal(o.bfs_grid(), s.bfs_grid())
    assert o.set_start(cfg.start) == s.set_start(cfg.start) == 1
    eo, go = _same_search(o, s, cfg.params.eps0, 4000, 4000)
    assert go["solved"] == 1


@pytest.fixture(scope="module")
def cfg2():
    return scenes.config2()


def test_config2_256cube_bfs_frontier_batch_and_bounded_search(cfg2):
    """cfg 2 at full size: BFS grid equality at 256^3, a B = 4096 frontier batch of real search states (the bench
    step's input), and the bounded eps 5 -> 1 search of the bench's planner leg."""
    from oracle_binding import Oracle
    from smpl_amd import capi
    _need_gpu()
    cfg = cfg2
    assert cfg.grid.dims == (256, 256, 256)
    o = Oracle(cfg)
    s = capi.Space.from_config(cfg, batch_states=4096)
    o.set_goal_joint(cfg.goal, cfg.goal_tol); s.set_goal_joint(cfg.goal, cfg.goal_tol)
    assert np.array_equal(o.goal_pose(), s.goal_pose())
    assert np.array_equal(o.bfs_grid(), s.bfs_grid())
    assert o.set_start(cfg.start) == s.set_start(cfg.start)
    eo, go = _same_search(o, s, cfg.params.eps0, 40000, 40000)
    assert go["gpu_batches"] > 0 and go["gpu_succ_evals"] >= go["committed_succ_evals"]
    # the frontier batch: the first 4096 states the search created, evaluated in one call; a sample of rows against the
    # oracle (its per-state evaluation is the slow side), every row against a second, smaller-batch evaluation
    Q = np.stack([s.get_state(i)[0] for i in range(1, 4097)])
    o.set_order(chain=True)
    got = s.expand_batch(Q)
    rows = np.r_[0:64, 2000:2064, 4032:4096]
    sub = _compare_batch(o, s, Q[rows])
    # the two calls may take different kernels (single launch / pipeline): equal wherever an output is defined
    # (include/smpl_amd.h: coord, h, cost on valid edges; q on every edge that is not inactive)
    assert np.array_equal(sub["flags"], got["flags"][rows])
    v = (sub["flags"] & 1) != 0
    act = (sub["flags"] & 0x10) == 0
    for k in ("coord", "h", "cost"):
        assert np.array_equal(sub[k][v], got[k][rows][v])
    assert np.array_equal(sub["q"][act], got["q"][rows][act])
    again = np.concatenate([s.expand_batch(Q[i:i + 512])["flags"] for i in range(0, 4096, 512)])
    assert np.array_equal(again, got["flags"])


def test_config4_shard_128_queries_on_one_gpu(cfg2):
    """The per-GPU shard of cfg 4: 128 independent (start, goal) queries of the seeded list on ONE shared grid and model
    through smplx_plan_multi.  Every query equals its solo GPU run; a sample of 8 equals the oracle."""
    from oracle_binding import Oracle
    from smpl_amd import capi
    _need_gpu()
    cfg = cfg2
    grid = capi.Grid(cfg.grid.origin, cfg.grid.dims, cfg.grid.res, cfg.grid.max_dist, cfg.grid.d2)
    model = capi.Model(cfg.robot_text)
    probe = capi.Space(model, grid, cfg.mprim, cfg.params, 256)
    cs, cg = scenes.config4_candidates()
    S, G = scenes.config4_queries(cs, cg, probe.state_valid_batch(cs)[0], probe.state_valid_batch(cg)[0])
    first, last = scenes.shard_range(0, 8)
    S, G = S[first:last], G[first:last]
    assert S.shape == (128, 7)
    nb = 1500
    spaces = []
    for a, b in zip(S, G):
        sp = capi.Space(model, grid, cfg.mprim, cfg.params, 1024)
        sp.set_goal_joint(b, cfg.goal_tol)
        sp.set_start(a)
        spaces.append(sp)
    multi, wall = capi.Space.plan_multi(spaces, 5.0, 1.0, 1.0, True, True, nb, nb, host_threads=4)
    assert len(multi) == 128 and wall > 0
    # solo runs of every query on a fresh space (one at a time: its own batches on its own stream)
    for i, (a, b) in enumerate(zip(S, G)):
        probe.set_goal_joint(b, cfg.goal_tol)
        probe.set_start(a)
        solo = probe.plan(5.0, 1.0, 1.0, True, True, nb, nb)
        m = multi[i]
        assert solo["solved"] == m["solved"] and solo["cost"] == m["cost"] and solo["expansions"] == m["expansions"], i
        assert np.array_equal(solo["expansion_log"], m["expansion_log"]), i
        assert np.array_equal(solo["path"], m["path"]), i
    for i in range(0, 128, 16):
        o = Oracle(cfg)
        o.set_goal_joint(G[i], cfg.goal_tol)
        o.set_start(S[i])
        o.search_params(5.0, 1.0, 1.0, True, True, nb, nb)
        e = o.plan()
        m = multi[i]
        assert e["ok"] == m["solved"] and e["cost"] == m["cost"] and np.array_equal(e["expansion_log"], m["expansion_log"]), i
    assert sum(m["expansions"] for m in multi) > 100 * nb


@pytest.fixture(scope="module")
def cfg5():
    return scenes.config5()


def test_config5_dual_arm_512cube(cfg5):
    """cfg 5 at full size: 14-DOF dual arm, 512^3 grid @ 0.01 m, 59 primitives, 85 checked link pairs: BFS grid,
    a 64-state expansion batch and a bounded eps 10 -> 1 search, all identical to the oracle."""
    from oracle_binding import Oracle
    from smpl_amd import capi
    _need_gpu()
    cfg = cfg5
    assert cfg.grid.dims == (512, 512, 512) and cfg.grid.res == 0.01
    # row N1 at this size: the field built on the GPU from the box list equals the host builder's
    gr = cfg.grid
    built = capi.Grid.from_boxes(gr.origin, gr.dims, gr.res, gr.max_dist, cfg.boxes)
    assert np.array_equal(built.d2(), gr.d2)
    del built
    s = capi.Space.from_config(cfg, batch_states=4096)
    assert (s.model.nvars, s.model.ntrees, s.model.npairs, s.M) == (14, 16, 85, 59)
    s.set_goal_joint(cfg.goal, cfg.goal_tol)
    o = Oracle(cfg)
    o.set_order(chain=True)
    o.set_goal_joint(cfg.goal, cfg.goal_tol)
    assert np.array_equal(o.bfs_grid(), s.bfs_grid())
    lim = scenes.ARM7_LIMITS + scenes.ARM7_LIMITS
    Q = np.vstack([np.array(cfg.start), scenes.random_states(lim, 63, 77)])
    got = _compare_batch(o, s, Q)
    assert (got["flags"] & 1).sum() > 50
    o.set_order(chain=False)
    assert o.set_start(cfg.start) == s.set_start(cfg.start)
    _same_search(o, s, cfg.params.eps0, 3000, 3000)


def test_config5_epsilon_decreasing_search_with_table_growth(cfg5, monkeypatch):
    """cfg 5 to its stated end as far as a test can afford ("eps-decreasing ARA* ... stress hash-table occupancy"):
    eps 10 -> 1 in steps of 1 with 20 000 expansions -- a solution at eps 10, then SEVEN epsilon steps (INCONS -> OPEN, reorder,
    re-expansions), 315 000 states in 64-byte table slots.  Twice on the GPU, both equal to the oracle: the host-driven loop
    with the device copy of the state table (2^18 slots, grown once past 131 072 states), and the device-resident search with
    a small first capacity (arena and table grown several times, the 512-thread block with the heap's LDS part at its smallest)."""
    from oracle_binding import Oracle
    from smpl_amd import capi
    _need_gpu()
    cfg = cfg5
    o = Oracle(cfg)
    o.set_goal_joint(cfg.goal, cfg.goal_tol)
    o.set_start(cfg.start)
    o.search_params(cfg.params.eps0, 1.0, 1.0, True, True, 20000, 20000)
    eo = o.plan()
    assert eo["ok"] == 1 and eo["cost"] > 0 and eo["eps"] <= cfg.params.eps0 - 2.0 and o.num_states() > (1 << 17)
    for mode in ("host", "device"):
        monkeypatch.setenv("SMPLX_SEARCH", mode)
        monkeypatch.setenv("SMPLX_DEVICE_TABLE", "1")
        s = capi.Space.from_config(cfg, batch_states=4096)
        if mode == "device":
            if not s.specialized()[0]:
                pytest.skip("generic kernels: the search kernel's block does not fit this robot's LDS scratch without the per-robot build")
            s.set_search_capacity(40000)
        s.set_goal_joint(cfg.goal, cfg.goal_tol)
        assert s.set_start(cfg.start) == 1
        go = s.plan(cfg.params.eps0, 1.0, 1.0, True, True, 20000, 20000)
        assert go["solved"] == 1 and go["cost"] == eo["cost"] and go["satisfied_eps"] == eo["eps"], mode
        assert go["expansions"] == eo["expansions"] and np.array_equal(go["expansion_log"], eo["expansion_log"]), mode
        assert np.array_equal(go["path"], eo["path"]) and s.num_states() == o.num_states(), mode
        assert go["committed_succ_evals"] == eo["succ_evals"], mode
        if mode == "device":
            sc = s.search_counters()
            assert sc["grows"] >= 3 and go["cache_misses"] == 0, sc
        del s


@pytest.fixture(scope="module")
def cfg2_solution(cfg2):
    """The oracle's cfg-2 search run to its first solution (eps 5, improve off): 194 806 expansions, ~17 s on one core."""
    from oracle_binding import Oracle
    o = Oracle(cfg2)
    o.set_goal_joint(cfg2.goal, cfg2.goal_tol)
    o.set_start(cfg2.start)
    o.search_params(cfg2.params.eps0, 1.0, 1.0, False, True, 1000000, 1000000)
    r = o.plan()
    r["num_states"] = o.num_states()
    return r


@pytest.mark.parametrize("mode", ["host", "device"])
def test_config2_eps5_search_reaches_its_goal(cfg2, cfg2_solution, mode, monkeypatch):
    """cfg 2 to its stated end ("eps = 5 ARA*", smpl_test/src/call_planner.cpp:1727-1729): the search runs until it reaches the
    goal -- 194 806 expansions, 1.09 M states -- and returns the oracle's path, cost and expansion log; once through the
    host-driven loop (what smplx_plan does for a lone query), once device-resident."""
    from smpl_amd import capi
    _need_gpu()
    cfg, eo = cfg2, cfg2_solution
    assert eo["ok"] == 1 and eo["cost"] > 0 and eo["expansions"] > 100000
    monkeypatch.setenv("SMPLX_SEARCH", mode)
    s = capi.Space.from_config(cfg, batch_states=4096)
    s.set_goal_joint(cfg.goal, cfg.goal_tol)
    assert s.set_start(cfg.start) == 1
    go = s.plan(cfg.params.eps0, 1.0, 1.0, False, True, 1000000, 1000000, cap=4096)
    assert go["solved"] == 1 and go["cost"] == eo["cost"] > 0 and go["satisfied_eps"] == eo["eps"] == 5.0
    assert go["expansions"] == eo["expansions"] and np.array_equal(go["expansion_log"], eo["expansion_log"])
    assert np.array_equal(go["path"], eo["path"]) and s.num_states() == eo["num_states"]
    assert go["committed_succ_evals"] == eo["succ_evals"]
    q = s.extract_path(go["path"])
    assert q.shape == (len(eo["path"]), 7) and np.array_equal(q[0], np.array(cfg.start))


def test_config4_shard_of_the_last_rank(cfg2, monkeypatch):
    """The cfg-4 shard of a non-zero rank: queries [896, 1024) of the seeded list (rank 7 of 8), one workgroup each in one
    launch of the device-resident search; every query equals the host-driven loop's result, a sample of 8 the oracle's."""
    from oracle_binding import Oracle
    from smpl_amd import capi
    _need_gpu()
    cfg = cfg2
    grid = capi.Grid(cfg.grid.origin, cfg.grid.dims, cfg.grid.res, cfg.grid.max_dist, cfg.grid.d2)
    model = capi.Model(cfg.robot_text)
    probe = capi.Space(model, grid, cfg.mprim, cfg.params, 256)
    cs, cg = scenes.config4_candidates()
    S, G = scenes.config4_queries(cs, cg, probe.state_valid_batch(cs)[0], probe.state_valid_batch(cg)[0])
    first, last = scenes.shard_range(7, 8)
    assert (first, last) == (896, 1024)
    S, G = S[first:last], G[first:last]
    nb = 2500

    def make():
        out = []
        for a, b in zip(S, G):
            sp = capi.Space(model, grid, cfg.mprim, cfg.params, 1024)
            sp.set_goal_joint(b, cfg.goal_tol)
            sp.set_start(a)
            out.append(sp)
        return out
    dev, wall = capi.Space.plan_multi(make(), 5.0, 1.0, 1.0, True, True, nb, nb)
    assert all(r["cache_misses"] == 0 for r in dev)               # the device-resident search ran
    monkeypatch.setenv("SMPLX_SEARCH", "host")
    host, _ = capi.Space.plan_multi(make(), 5.0, 1.0, 1.0, True, True, nb, nb, host_threads=8)
    for i, (a, b) in enumerate(zip(dev, host)):
        assert a["solved"] == b["solved"] and a["cost"] == b["cost"] and a["expansions"] == b["expansions"], i
        assert np.array_equal(a["expansion_log"], b["expansion_log"]) and np.array_equal(a["path"], b["path"]), i
    for i in range(0, 128, 16):
        o = Oracle(cfg)
        o.set_goal_joint(G[i], cfg.goal_tol)
        o.set_start(S[i])
        o.search_params(5.0, 1.0, 1.0, True, True, nb, nb)
        e = o.plan()
        m = dev[i]
        assert e["ok"] == m["solved"] and e["cost"] == m["cost"] and np.array_equal(e["expansion_log"], m["expansion_log"]), i
    assert sum(m["solved"] for m in dev) > 0


def test_strictly_growing_and_shrinking_batches_on_one_space(small_cfg):
    """Regression for the k_pipe_configs memory fault of round 1: one space, the four-kernel
    pipeline forced for every size, batches that grow past every earlier reservation (buffers are re-allocated and
    re-carved) and shrink again; each batch compared with the oracle row by row."""
    from oracle_binding import Oracle
    from smpl_amd import capi
    _need_gpu()
    cfg = small_cfg
    o = Oracle(cfg)
    o.set_order(chain=True)
    s = capi.Space.from_config(cfg, no_small_kernel=True)
    o.set_goal_joint(cfg.goal, cfg.goal_tol); s.set_goal_joint(cfg.goal, cfg.goal_tol)
    Q = scenes.random_states(scenes.ARM7_LIMITS, 1200, 91)
    ref = {}
    for B in [1, 2, 3, 7, 64, 65, 129, 300, 777, 1200, 5, 640, 1, 1200]:
        got = s.expand_batch(Q[:B])
        for i in range(0, B, max(1, B // 24)):
            if i not in ref:
                ref[i] = o.eval_state(Q[i])
            e = ref[i]
            assert np.array_equal(e["flags"], got["flags"][i]), (B, i)
            v = (e["flags"] & 1) != 0
            assert np.array_equal(e["coord"][v], got["coord"][i][v]) and np.array_equal(e["h"][v], got["h"][i][v])
    # the same through the planner (its batch sizes follow OPEN: 1, 2, ... up to batch_states) with the pipeline forced
    for bs in (7, 64, 1024):
        sp = capi.Space.from_config(cfg, batch_states=bs, no_small_kernel=True)
        sp.set_goal_joint(cfg.goal, cfg.goal_tol)
        sp.set_start(cfg.start)
        oo = Oracle(cfg)
        oo.set_goal_joint(cfg.goal, cfg.goal_tol)
        oo.set_start(cfg.start)
        oo.search_params(5.0, 1.0, 1.0, True, True, 2000, 2000)
        e = oo.plan()
        g = sp.plan(5.0, 1.0, 1.0, True, True, 2000, 2000)
        assert e["cost"] == g["cost"] and np.array_equal(e["expansion_log"], g["expansion_log"]), bs


def test_non_finite_and_absurd_inputs_are_refused(small_cfg):
    """A NaN / inf / 1e300 joint value must come back as SMPLX_E_ARG, not reach the device (the reference's limit
    folding, kdl_robot_model.cpp:210-235, would not terminate on it)."""
    from smpl_amd import capi
    _need_gpu()
    cfg = small_cfg
    s = capi.Space.from_config(cfg)
    s.set_goal_joint(cfg.goal, cfg.goal_tol)
    for bad in (np.nan, np.inf, -np.inf, 1e300):
        q = np.array([cfg.start, cfg.start]); q[1, 2] = bad
        for call in (lambda: s.state_valid_batch(q), lambda: s.edge_valid_batch(q, q), lambda: s.expand_batch(q),
                     lambda: s.heuristic_batch(q), lambda: s.set_start(q[1]), lambda: s.sphere_positions(q)):
            with pytest.raises(capi.SmplxError) as err:
                call()
            assert err.value.code == -1
    assert s.state_valid_batch(np.array([cfg.start]))[0][0] == 1      # the space is still usable


def test_k5_device_state_table_and_compacted_successor_stream(small_cfg):
    """K5 (manip_lattice.cpp:1302-1354 on the device): after a search the device copy of the state table names every
    committed coordinate; the compact stream built with wavefront ballots lists exactly the valid successors in
    (state, primitive) order, with full records for the unknown ones and the goal successors."""
    import struct
    from smpl_amd import capi
    _need_gpu()
    cfg = small_cfg
    s = capi.Space.from_config(cfg, batch_states=256)
    s.set_goal_joint(cfg.goal, cfg.goal_tol)
    s.set_start(cfg.start)
    r = s.plan(5.0, 1.0, 1.0, True, True, 1500, 1500)
    assert r["expansions"] == 1500
    n = s.num_states()
    host = {tuple(s.get_state(i)[1]): i for i in range(1, n)}      # the committed table: coordinate -> id
    assert len(host) == n - 1
    s.table_sync()
    g = np.array(cfg.goal)
    near = np.array([g + np.array([0, 0, 0, 0, 4, 0, 0]) * DEG, g - np.array([0, 0, 0, 0, 0, 4, 0]) * DEG])
    Q = np.vstack([np.stack([s.get_state(i)[0] for i in range(1, 301)]), near,
                   scenes.random_states(scenes.ARM7_LIMITS, 60, 17)])
    B, M, N = Q.shape[0], s.M, s.N
    dense = s.expand_batch(Q)
    got = s.expand_batch_k5(Q)
    for k in ("flags", "coord", "q", "h"):
        assert np.array_equal(dense[k], got[k]), k
    valid = (got["flags"] & 1) != 0
    goal = (got["flags"] & 2) != 0
    want_id = np.full((B, M), -1, np.int32)
    for i, p in zip(*np.nonzero(valid)):
        want_id[i, p] = host.get(tuple(got["coord"][i, p]), -1)
    assert np.array_equal(got["succ_id"], want_id)
    assert (want_id[valid] >= 0).sum() > 500 and (want_id[valid] < 0).sum() > 200 and goal.sum() >= 2
    # the compact stream
    tot = got["totals"]
    assert tot[2] == 0 and tot[0] == valid.sum()
    need_b = valid & ((want_id < 0) | goal)
    assert tot[1] == need_b.sum()
    ints = (N + 2) // 2 * 2
    assert s.compact_rec_b_bytes() == ints * 4 + N * 8
    seq_a, seq_b = [], []
    for ba, ca, bb, cb in got["block_tab"]:
        seq_a += [tuple(x) for x in got["rec_a"][ba:ba + ca]]
        seq_b += [bytes(x) for x in got["rec_b"][bb:bb + cb]]
    assert len(seq_a) == tot[0] and len(seq_b) == tot[1]
    ib = 0
    for (i, p), (rid, meta) in zip(zip(*np.nonzero(valid)), seq_a):      # np.nonzero is (state, primitive) order
        assert rid == want_id[i, p]
        assert meta == (p | (0x100 if goal[i, p] else 0) | (i << 9))
        if need_b[i, p]:
            rec = seq_b[ib]; ib += 1
            vals = struct.unpack(f"<{ints}i{N}d", rec)
            assert vals[0] == got["h"][i, p]
            assert list(vals[1:1 + N]) == list(got["coord"][i, p])
            assert list(vals[ints:]) == list(got["q"][i, p])
    assert ib == len(seq_b)
    # a region that is too small: flagged, dense outputs unaffected
    small = s.expand_batch_k5(Q, cap_a=64, cap_b=64)
    assert small["totals"][2] == 1 and np.array_equal(small["flags"], dense["flags"]) and np.array_equal(small["succ_id"], want_id)


def test_device_table_on_and_off_give_the_same_search(small_cfg, monkeypatch):
    """The ids the device table hands back are only a shortcut for the host's own getOrCreateState: the search is the
    same with SMPLX_DEVICE_TABLE=0 (every lookup on the host) -- and the table survives growing past its first size."""
    from smpl_amd import capi
    _need_gpu()
    cfg = small_cfg
    runs = []
    monkeypatch.setenv("SMPLX_SEARCH", "host")      # (the device-resident search always keeps the table on the device)
    for env in ("1", "0"):
        monkeypatch.setenv("SMPLX_DEVICE_TABLE", env)
        s = capi.Space.from_config(cfg, batch_states=512)
        s.set_goal_joint(cfg.goal, cfg.goal_tol)
        s.set_start(cfg.start)
        runs.append((s.plan(5.0, 1.0, 1.0, True, True, 60000, 60000), s.num_states()))
    (a, na), (b, nb) = runs
    assert na == nb and na > (1 << 17)                      # more states than half the initial 2^18 slots: it grew
    assert a["cost"] == b["cost"] and np.array_equal(a["expansion_log"], b["expansion_log"]) and np.array_equal(a["path"], b["path"])


PR2_RIGHT_ARM_LIMITS = [(-2.1353981634, 0.564601836603), (-0.3536, 1.2963), (-3.75, 0.65), (-2.1213, -0.15),
                        (-np.pi, np.pi), (-2.0, -0.1), (-np.pi, np.pi)]


def test_config3_pr2_right_arm_as_data(cfg3_pr2):
    """SURVEY cfg 3 as specified: the PR2 right arm from data files (the reference's collision_model_pr2.yaml, a URDF
    subset of the arm, the right-arm rows of the demo's allowed-collision matrix) in the 150^3 cluttered-tabletop scene,
    eps 100.  BFS grid, an expansion batch (start + random states within the PR2's limits), random state validity and the
    bounded search against the oracle fed the same model text."""
    from oracle_binding import Oracle
    from smpl_amd import capi
    _need_gpu()
    cfg = cfg3_pr2
    o = Oracle(cfg)
    s = capi.Space.from_config(cfg, batch_states=1024)
    o.set_goal_joint(cfg.goal, cfg.goal_tol); s.set_goal_joint(cfg.goal, cfg.goal_tol)
    assert np.array_equal(o.goal_pose(), s.goal_pose())
    assert np.array_equal(o.bfs_grid(), s.bfs_grid())
    Q = np.vstack([np.array(cfg.start), scenes.random_states(PR2_RIGHT_ARM_LIMITS, 95, 33)])
    o.set_order(chain=True)
    _compare_batch(o, s, Q)
    R = scenes.random_states(PR2_RIGHT_ARM_LIMITS, 2000, 34)
    ok, lk = s.state_valid_batch(R)
    want = np.array([o.state_valid(q)[0] for q in R[:400]])
    assert np.array_equal(ok[:400].astype(bool), want.astype(bool)) and 0.05 < ok.mean() < 0.95
    o.set_order(chain=False)
    assert o.set_start(cfg.start) == s.set_start(cfg.start) == 1
    eo, go = _same_search(o, s, 100.0, 2500, 2500)
    assert go["solved"] == 1 and go["cost"] > 0


@pytest.mark.parametrize("threads,no_small", [(1, False), (3, False), (3, True), (8, False)])
def test_multi_query_driver_scheduling_does_not_change_results(small_cfg, monkeypatch, threads, no_small):
    """How the host-driven multi-query driver schedules its batches (one thread in rounds, worker threads + one GPU
    submitter, single-launch zero-copy kernel or the four-kernel pipeline) decides WHEN and HOW batches are evaluated,
    never what a query computes: eight queries through smplx_plan_multi under each setting equal their solo runs."""
    from smpl_amd import capi
    _need_gpu()
    cfg = small_cfg
    DEG = np.pi / 180.0
    cells = [[-49, 7, 21, -14, -8, -12, 16], [-21, 7, 14, -7, 8, -4, 12], [-35, 14, 7, -14, 4, -8, 8], [-42, 10, 14, -10, 0, -8, 12],
             [-28, 7, 7, -7, 4, -4, 4], [-14, 14, 21, -21, -8, 0, 8], [-56, 0, 14, -7, 0, -12, 16], [-7, 7, 28, -14, 8, -8, 0]]
    goals = [[cfg.start[i] + c * DEG for i, c in enumerate(cs)] for cs in cells]
    grid = capi.Grid(cfg.grid.origin, cfg.grid.dims, cfg.grid.res, cfg.grid.max_dist, cfg.grid.d2)
    model = capi.Model(cfg.robot_text)

    def make():
        out = []
        for g in goals:
            sp = capi.Space(model, grid, cfg.mprim, cfg.params, 512, no_small_kernel=no_small)
            sp.set_goal_joint(g, cfg.goal_tol); sp.set_start(cfg.start)
            out.append(sp)
        return out
    monkeypatch.setenv("SMPLX_SEARCH", "host")
    solo = [sp.plan(5.0, 1.0, 1.0, True, True, 3000, 2000) for sp in make()]
    multi, _ = capi.Space.plan_multi(make(), 5.0, 1.0, 1.0, True, True, 3000, 2000, host_threads=threads)
    for a, b in zip(solo, multi):
        assert a["solved"] == b["solved"] and a["cost"] == b["cost"] and np.array_equal(a["expansion_log"], b["expansion_log"])
        assert np.array_equal(a["path"], b["path"]) and a["committed_succ_evals"] == b["committed_succ_evals"]


