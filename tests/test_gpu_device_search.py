"""SURVEY row N2 -- the device-resident ARA* (smpl_amd/csrc/search_kernel.h, search_host.h): one persistent workgroup per
query runs pop -> expand -> getOrCreateState -> push on the GPU.  Compared through the C-ABI with
  * the reference's own intrusive_heap (tests/golden/heap_ref.json, produced by its header compiled in place),
  * the oracle's sequential ARA* on the same seeded inputs: expansion log (= expanded-state set and order), state ids,
    joint values and coordinates of every state, path, cost, satisfied epsilon, successor-evaluation count,
  * the host-driven loop of the same engine (SMPLX_SEARCH=host), which is what an SBPL-shaped caller gets.
Parity status: the oracle is pinned by the reference only for the heap."""
import json
import os

import numpy as np
import pytest

from smpl_amd import scenes

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "heap_ref.json")


def _need_gpu():
    from smpl_amd import capi
    if capi.lib().smplx_device_count() == 0:
        pytest.fail("no GPU visible: the gpu-marked tests must run on the MI355X box")


def _heap_ops(ops, lds_entries):
    import ctypes as C
    from smpl_amd import capi
    ops = np.ascontiguousarray(ops, np.int32).reshape(-1, 2)
    out = np.zeros(ops.shape[0], np.int32)
    L = capi.lib()
    L.smplx_test_heap_ops.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    rc = L.smplx_test_heap_ops(ops.ctypes.data, ops.shape[0], lds_entries, out.ctypes.data)
    assert rc == 0, L.smplx_last_error().decode()
    return out


@pytest.mark.parametrize("lds_entries", [1, 5, 64, 4096])
def test_device_heap_matches_the_reference_heap_golden(lds_entries):
    """push / pop / decrease / increase / erase / make of the search kernel's heap against the top-after-each-op record of
    smpl/include/smpl/intrusive_heap.h, with the LDS / HBM boundary of the heap array at several places (lds_entries = 1:
    the whole heap in HBM)."""
    _need_gpu()
    cases = json.load(open(GOLD))["cases"]
    assert len(cases) >= 6
    for c in cases:
        got = _heap_ops(c["ops"], lds_entries)
        assert got.tolist() == c["top_after_each_op"], f"case seed={c['seed']}"
    # priorities {5,3,3,9,1} pop in element order 4,1,2,0,3 (ties: the newer element sifts above)
    ops = [(0, 5), (0, 3), (0, 3), (0, 9), (0, 1)] + [(1, 0)] * 5
    assert _heap_ops(ops, lds_entries).tolist()[4:] == [4, 1, 2, 0, 3, -1]


def test_device_heap_long_random_sequence_against_the_oracle_heap(small_cfg):
    """A few thousand random ops (the make() of op 4 runs level-parallel on the device) against the oracle's IntrusiveHeap,
    which tests/test_heap_golden.py pins to the reference."""
    from oracle_binding import Oracle
    _need_gpu()
    o = Oracle(small_cfg)
    rng = np.random.default_rng(5)
    for lds_entries in (7, 256):
        ops, n = [], 0
        for _ in range(6000):
            r = rng.random()
            if r < 0.55 or n == 0:
                ops.append((0, int(rng.integers(0, 40)))); n += 1
            elif r < 0.8:
                ops.append((1, 0))
            elif r < 0.9:
                ops.append((2, (int(rng.integers(0, min(n, 2047))) << 20) | int(rng.integers(0, 20))))
            elif r < 0.95:
                ops.append((3, int(rng.integers(0, n))))
            elif r < 0.99:
                ops.append((5, (int(rng.integers(0, min(n, 2047))) << 20) | int(rng.integers(20, 60))))
            else:
                ops.append((4, 0))
        want = o.heap_run(np.array(ops, np.int32))
        assert np.array_equal(_heap_ops(ops, lds_entries), want)


def _check_against_oracle(o, s, eo, go):
    assert eo["ok"] == go["solved"] and eo["expansions"] == go["expansions"]
    assert np.array_equal(eo["expansion_log"], go["expansion_log"])       # expanded-state set and order, ids bit-exact
    assert eo["cost"] == go["cost"] and np.array_equal(eo["path"], go["path"])
    assert eo["eps"] == go["satisfied_eps"]
    assert o.num_states() == s.num_states()
    assert eo["succ_evals"] == go["committed_succ_evals"]


@pytest.mark.parametrize("semantics", ["fork", "upstream"])
@pytest.mark.parametrize("goal_kind", ["joint", "xyz"])
def test_device_search_equals_the_oracle(small_cfg, goal_kind, semantics, monkeypatch):
    """config_small, both primitive semantics, joint and XYZ goals: log, ids, cost, path, epsilon, evaluation count equal to
    the oracle's; every state the device created has the oracle's joint values, coordinate and heuristic; the committed
    successor lists (served to a later GetSuccs caller) are the oracle's; the path's joint values equal the host loop's."""
    from oracle_binding import Oracle
    from smpl_amd import capi
    _need_gpu()
    monkeypatch.setenv("SMPLX_SEARCH", "device")      # fail instead of silently taking the host loop
    cfg = small_cfg
    rot = semantics == "fork"
    o = Oracle(cfg, xy_rotate=rot)
    s = capi.Space.from_config(cfg, batch_states=256, xy_rotate=rot)
    if goal_kind == "joint":
        o.set_goal_joint(cfg.goal, cfg.goal_tol); s.set_goal_joint(cfg.goal, cfg.goal_tol)
    else:
        p = o.planning_fk(cfg.goal)
        o.set_goal_xyz(p, [0.04] * 3); s.set_goal_xyz(p, [0.04] * 3)
    assert o.set_start(cfg.start) == s.set_start(cfg.start)
    o.search_params(5.0, 1.0, 1.0, True, True, 6000, 3000)
    eo = o.plan()
    go = s.plan(5.0, 1.0, 1.0, True, True, 6000, 3000)
    _check_against_oracle(o, s, eo, go)
    assert go["cache_misses"] == 0 and go["gpu_batches"] >= 1
    sc = s.search_counters()
    assert sc["searches"] == 1 and sc["device_states"] == o.num_states() and sc["heap_cache_entries"] >= 64
    n = s.num_states()
    for i in list(range(1, 400)) + list(range(400, n, 97)) + [n - 1]:
        eq, ecd = o.get_state(i)
        gq, gcd = s.get_state(i)
        assert np.array_equal(eq, gq) and np.array_equal(ecd, gcd), i
        assert o.heuristic_q(eq) == s.goal_heuristic(i), i
    for i in [int(x) for x in eo["expansion_log"][:200]] + [int(eo["expansion_log"][-1])]:
        es, ec = o.get_succs(i)
        gs, gc = s.get_succs(i)
        assert np.array_equal(es, gs) and np.array_equal(ec, gc), i
    if go["solved"]:
        q = s.extract_path(go["path"])
        assert np.array_equal(q[0], np.array(cfg.start))
        monkeypatch.setenv("SMPLX_SEARCH", "host")
        h = capi.Space.from_config(cfg, batch_states=256, xy_rotate=rot)
        if goal_kind == "joint":
            h.set_goal_joint(cfg.goal, cfg.goal_tol)
        else:
            h.set_goal_xyz(o.planning_fk(cfg.goal), [0.04] * 3)
        h.set_start(cfg.start)
        gh = h.plan(5.0, 1.0, 1.0, True, True, 6000, 3000)
        assert np.array_equal(gh["path"], go["path"]) and gh["cache_misses"] > 0
        assert np.array_equal(h.extract_path(gh["path"]), q)


@pytest.mark.parametrize("helper_wave", [True, False])
def test_device_search_through_epsilon_steps(small_cfg, monkeypatch, helper_wave):
    """eps 5 -> 1 in steps of 1 on config_small with room for several improvement rounds: INCONS -> OPEN, the recomputation
    of f and the level-parallel make() of every epsilon step, re-expansions served from the committed lists.  Every
    expansion of every round, the cost after the last finished round and its epsilon agree with the oracle.  Once with the
    helper wave that does the successors' bookkeeping beside the search wave (the default where the block has room), once
    without (test hook: the search wave does it inline, as for robots with many primitives)."""
    from oracle_binding import Oracle
    from smpl_amd import capi
    _need_gpu()
    monkeypatch.setenv("SMPLX_SEARCH", "device")
    cfg = small_cfg
    o = Oracle(cfg)
    s = capi.Space.from_config(cfg, batch_states=256)
    s.set_search_helper(helper_wave)
    o.set_goal_joint(cfg.goal, cfg.goal_tol); s.set_goal_joint(cfg.goal, cfg.goal_tol)
    assert o.set_start(cfg.start) == s.set_start(cfg.start)
    o.search_params(5.0, 1.0, 1.0, True, True, 20000, 60000)
    eo = o.plan()
    go = s.plan(5.0, 1.0, 1.0, True, True, 20000, 60000)
    _check_against_oracle(o, s, eo, go)
    assert go["solved"] == 1 and go["satisfied_eps"] <= 3.0 and go["cost"] > 0      # at least two epsilon steps finished
    assert go["expansions"] > go["expansions_init"] > 0


def test_device_search_outgrows_its_buffers(small_cfg, monkeypatch):
    """A first capacity of 2 000 states (test hook): the workgroup stops for room (SMPLX_SS_GROW) many times, the host
    enlarges arena and state table and launches again; same search as the oracle's."""
    from oracle_binding import Oracle
    from smpl_amd import capi
    _need_gpu()
    monkeypatch.setenv("SMPLX_SEARCH", "device")
    cfg = small_cfg
    o = Oracle(cfg)
    s = capi.Space.from_config(cfg, batch_states=256)
    s.set_search_capacity(2000)
    o.set_goal_joint(cfg.goal, cfg.goal_tol); s.set_goal_joint(cfg.goal, cfg.goal_tol)
    assert o.set_start(cfg.start) == s.set_start(cfg.start)
    o.search_params(5.0, 1.0, 1.0, True, True, 12000, 6000)
    eo = o.plan()
    go = s.plan(5.0, 1.0, 1.0, True, True, 12000, 6000)
    _check_against_oracle(o, s, eo, go)
    assert s.search_counters()["grows"] >= 4


def test_device_and_host_searches_share_one_lattice(small_cfg, monkeypatch):
    """One space: a device search, then a host-driven search from another start on the same goal (it must see every state
    the device created, with the device's ids), then a device search again (it must see the host's).  Each equals the
    oracle run with the same sequence of calls."""
    from oracle_binding import Oracle
    from smpl_amd import capi
    _need_gpu()
    cfg = small_cfg
    DEG = np.pi / 180.0
    o = Oracle(cfg)
    s = capi.Space.from_config(cfg, batch_states=256)
    o.set_goal_joint(cfg.goal, cfg.goal_tol); s.set_goal_joint(cfg.goal, cfg.goal_tol)
    starts = [np.array(cfg.start), np.array(cfg.start) + np.array([7, 0, 7, 0, 0, 0, 4]) * DEG,
              np.array(cfg.start) + np.array([-14, 7, 0, -7, 4, 0, 0]) * DEG]
    for k, (st, mode) in enumerate(zip(starts, ["device", "host", "device"])):
        monkeypatch.setenv("SMPLX_SEARCH", mode)
        assert o.set_start(st) == s.set_start(st)
        o.search_params(5.0, 1.0, 1.0, True, True, 2500, 1500)
        eo = o.plan()
        go = s.plan(5.0, 1.0, 1.0, True, True, 2500, 1500)
        _check_against_oracle(o, s, eo, go)
        assert (go["cache_misses"] == 0) == (mode == "device"), k


def test_config2_bounded_search_on_the_device(monkeypatch):
    """BASELINE cfg 2 (256^3, eps 5): a bounded search, device-resident, against the oracle."""
    from oracle_binding import Oracle
    from smpl_amd import capi
    _need_gpu()
    monkeypatch.setenv("SMPLX_SEARCH", "device")
    cfg = scenes.config2()
    o = Oracle(cfg)
    s = capi.Space.from_config(cfg, batch_states=4096)
    o.set_goal_joint(cfg.goal, cfg.goal_tol); s.set_goal_joint(cfg.goal, cfg.goal_tol)
    assert o.set_start(cfg.start) == s.set_start(cfg.start)
    o.search_params(cfg.params.eps0, 1.0, 1.0, True, True, 40000, 40000)
    eo = o.plan()
    go = s.plan(cfg.params.eps0, 1.0, 1.0, True, True, 40000, 40000)
    _check_against_oracle(o, s, eo, go)
    assert go["cache_misses"] == 0


def test_shard_of_queries_in_one_launch(small_cfg, monkeypatch):
    """Sixteen queries that share grid, robot and primitives: one workgroup each in ONE launch of k_search.  Every query
    equals its solo device run and the host-driven loop."""
    from smpl_amd import capi
    _need_gpu()
    cfg = small_cfg
    DEG = np.pi / 180.0
    rng = np.random.default_rng(3)
    cells = rng.integers(-6, 7, size=(16, 7)) * np.array([7, 7, 7, 7, 4, 4, 4])
    goals = [[cfg.start[i] + c * DEG for i, c in enumerate(cs)] for cs in cells]
    grid = capi.Grid(cfg.grid.origin, cfg.grid.dims, cfg.grid.res, cfg.grid.max_dist, cfg.grid.d2)
    model = capi.Model(cfg.robot_text)
    probe = capi.Space(model, grid, cfg.mprim, cfg.params, 256)
    ok = probe.state_valid_batch(np.array(goals))[0].astype(bool)
    goals = [g for g, k in zip(goals, ok) if k]
    assert len(goals) >= 8

    def make():
        out = []
        for g in goals:
            sp = capi.Space(model, grid, cfg.mprim, cfg.params, 512)
            sp.set_goal_joint(g, cfg.goal_tol); sp.set_start(cfg.start)
            out.append(sp)
        return out
    monkeypatch.setenv("SMPLX_SEARCH", "device")
    multi, wall = capi.Space.plan_multi(make(), 5.0, 1.0, 1.0, True, True, 3000, 2000)
    solo = [sp.plan(5.0, 1.0, 1.0, True, True, 3000, 2000) for sp in make()]
    monkeypatch.setenv("SMPLX_SEARCH", "host")
    host, _ = capi.Space.plan_multi(make(), 5.0, 1.0, 1.0, True, True, 3000, 2000, host_threads=3)
    for a, b, c in zip(solo, multi, host):
        for x in (b, c):
            assert a["solved"] == x["solved"] and a["cost"] == x["cost"] and np.array_equal(a["expansion_log"], x["expansion_log"])
            assert np.array_equal(a["path"], x["path"]) and a["committed_succ_evals"] == x["committed_succ_evals"]
    assert all(m["cache_misses"] == 0 for m in multi) and any(h["cache_misses"] > 0 for h in host)
