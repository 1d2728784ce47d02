"""Row N1: distance-field construction on the GPU (smpl_amd/csrc/field.hip) -- exact Euclidean transform with the
reference's cap and border rule, checked against a brute-force nearest-obstacle search, against the host builder of
the test scenes at BASELINE sizes, and through a planner query that runs on a GPU-built grid."""
import numpy as np
import pytest

from smpl_amd import scenes

pytestmark = pytest.mark.gpu


def _brute_force(occ, dmax):
    """Squared distance of every interior cell to the nearest occupied or border cell, capped (O(cells x obstacles))."""
    nx, ny, nz = occ.shape
    pad = np.ones((nx + 2, ny + 2, nz + 2), bool)
    pad[1:-1, 1:-1, 1:-1] = occ
    obs = np.argwhere(pad).astype(np.int64) - 1          # interior coordinates; the border layer sits at -1 and n
    out = np.zeros(occ.shape, np.int64)
    cells = np.argwhere(np.ones(occ.shape, bool)).astype(np.int64)
    for k in range(0, cells.shape[0], 2048):
        c = cells[k:k + 2048]
        d = ((c[:, None, :] - obs[None, :, :]) ** 2).sum(axis=2).min(axis=1)
        out[c[:, 0], c[:, 1], c[:, 2]] = d
    return np.minimum(out, dmax * dmax).astype(np.int32)


def test_field_equals_bruteforce_on_small_grids():
    from smpl_amd import capi
    rng = np.random.default_rng(5)
    for dims, res, max_dist in [((12, 9, 7), 0.05, 0.2), ((16, 16, 16), 0.02, 0.4), ((5, 21, 8), 0.1, 0.35)]:
        origin = (-0.3, 0.1, 0.0)
        g = capi.Grid.empty(origin, dims, res, max_dist)
        dmax = int(np.ceil(max_dist * (1.0 / res)))
        occ = np.zeros(dims, bool)
        assert np.array_equal(g.d2(), _brute_force(occ, dmax))          # empty grid: distance to the border layer
        # a handful of occupied cells through addPointsToField (cell centres), plus points outside the grid (skipped)
        cells = np.unique(rng.integers(0, dims, size=(9, 3)), axis=0)
        pts = np.asarray(origin) + cells * res       # gridToWorld (distance_map.hpp:506-518): cell centres at origin + res * cell
        assert np.array_equal(scenes.world_to_grid(origin, res, pts), cells)
        g.add_points(np.vstack([pts, [[9.0, 9.0, 9.0], [-5.0, 0.0, 0.0]]]))
        occ[cells[:, 0], cells[:, 1], cells[:, 2]] = True
        assert np.array_equal(g.d2(), _brute_force(occ, dmax))
        # removePointsFromField
        g.remove_points(pts[:4])
        occ[cells[:4, 0], cells[:4, 1], cells[:4, 2]] = False
        assert np.array_equal(g.d2(), _brute_force(occ, dmax))


def test_field_equals_host_builder_at_config_sizes():
    """cfg 2 (256^3, 65 boxes, cap 20 cells) and cfg 3 (150^3: not a multiple of the brick, cap 90 cells)."""
    from smpl_amd import capi
    for cfg in (scenes.config2(), scenes.config3()):
        gr = cfg.grid
        g = capi.Grid.from_boxes(gr.origin, gr.dims, gr.res, gr.max_dist, cfg.boxes)
        assert np.array_equal(g.d2(), gr.d2), cfg.name


def test_planner_on_a_gpu_built_grid_equals_the_oracle(small_cfg):
    """A .env-style box list goes to the GPU, the field is built there, and the search that runs on it is the oracle's
    (the oracle is handed the field the GPU built)."""
    import copy
    from oracle_binding import Oracle
    from smpl_amd import capi
    cfg = copy.copy(small_cfg)
    gr = cfg.grid
    g = capi.Grid.from_boxes(gr.origin, gr.dims, gr.res, gr.max_dist, cfg.boxes)
    built = g.d2()
    assert np.array_equal(built, gr.d2)
    cfg.grid = scenes.Grid(gr.origin, gr.dims, gr.res, gr.max_dist, built)
    s = capi.Space(capi.Model(cfg.robot_text), g, cfg.mprim, cfg.params, 256)
    s.set_goal_joint(cfg.goal, cfg.goal_tol)
    s.set_start(cfg.start)
    o = Oracle(cfg)
    o.set_goal_joint(cfg.goal, cfg.goal_tol)
    o.set_start(cfg.start)
    o.search_params(5.0, 1.0, 1.0, True, True, 2000, 2000)
    e = o.plan()
    r = s.plan(5.0, 1.0, 1.0, True, True, 2000, 2000)
    assert e["cost"] == r["cost"] and np.array_equal(e["expansion_log"], r["expansion_log"])
    # a grid that came from a finished field cannot be edited
    fixed = capi.Grid(gr.origin, gr.dims, gr.res, gr.max_dist, gr.d2)
    with pytest.raises(capi.SmplxError):
        fixed.add_points(np.zeros((1, 3)))


def test_reference_env_file_to_gpu_grid():
    """Row N4 wired to the C-ABI: the reference's own smpl_test/env/tabletop.env (tests/golden) is parsed as
    call_planner.cpp:158-207 parses it and its boxes become the voxel grid on the GPU; the field equals config 1's."""
    import os
    from smpl_amd import capi, formats
    env = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tabletop.env")).read()
    objs = formats.parse_env(env)
    cfg = scenes.config1()
    g = capi.Grid.from_boxes(cfg.grid.origin, cfg.grid.dims, cfg.grid.res, cfg.grid.max_dist, [(c, s) for (_, c, s) in objs])
    assert np.array_equal(g.d2(), cfg.grid.d2)


def test_reference_counts_duplicate_adds_and_partial_removes():
    """OccupancyGrid's ref_counted mode (occupancy_grid.cpp:357-406, 424-441) against a host model of its loops: points listed
    twice count twice, a cell is an obstacle while its count is positive, removing from an empty cell does nothing; the
    field after every step equals the brute-force transform of the cells with a positive count."""
    from smpl_amd import capi
    rng = np.random.default_rng(11)
    dims, res, max_dist, origin = (14, 11, 9), 0.05, 0.25, (0.2, -0.1, 0.0)
    dmax = int(np.ceil(max_dist * (1.0 / res)))
    g = capi.Grid.empty(origin, dims, res, max_dist)
    g.add_points(np.asarray(origin) + np.array([[3, 3, 3], [4, 3, 3]]) * res)      # obstacles from before the counts are switched on
    g.set_ref_counted(True)
    counts = np.zeros(dims, np.int64)
    counts[3, 3, 3] = counts[4, 3, 3] = 1                                          # initRefCounts: 1 where occupied
    assert np.array_equal(g.counts(), counts)

    def pts(cells):
        return np.asarray(origin) + np.asarray(cells) * res

    for step in range(8):
        cells = rng.integers(0, dims, size=(12, 3))
        cells = np.vstack([cells, cells[:5], cells[:2], [[3, 3, 3]]])             # duplicates inside one call
        outside = np.array([[40.0, 0.0, 0.0], [0.0, -7.0, 0.0]])
        if step % 2 == 0:
            g.add_points(np.vstack([pts(cells), outside]))
            for c in cells:
                counts[tuple(c)] += 1
        else:
            rem = np.vstack([cells[::2], rng.integers(0, dims, size=(6, 3))])     # partial removal + cells that were never added
            g.remove_points(np.vstack([pts(rem), outside]))
            for c in rem:
                if counts[tuple(c)] > 0:
                    counts[tuple(c)] -= 1
        assert np.array_equal(g.counts(), counts), step
        assert np.array_equal(g.d2(), _brute_force(counts > 0, dmax)), step
    assert (counts > 1).any() and (counts == 0).any()


def test_incremental_edit_equals_a_field_built_from_scratch():
    """An edit recomputes only the window its cells can reach (their bounding box grown by dmax); after a series of adds,
    removes and updatePointsInField the field equals one built from scratch from the same obstacles -- at 128^3 with a
    cap of 20 cells, where the window of a point is 41^3 of 2 M cells."""
    from smpl_amd import capi
    rng = np.random.default_rng(3)
    dims, res, max_dist, origin = (128, 128, 128), 0.02, 0.4, (-0.5, -0.5, 0.0)
    boxes = [((0.3, 0.2, 0.6), (0.4, 0.6, 0.04)), ((-0.1, 0.9, 1.4), (0.2, 0.2, 0.3))]
    g = capi.Grid.from_boxes(origin, dims, res, max_dist, boxes)
    whole = 128 ** 3

    def world(cells):
        return np.asarray(origin) + np.asarray(cells, float) * res

    # one point: a 41^3 window (clipped at the faces)
    g.add_points(world([[64, 64, 64]]))
    assert g.last_edit_cells() == 41 ** 3
    g.add_points(world([[2, 120, 64]]))
    assert g.last_edit_cells() == 23 * 28 * 41
    added = [[64, 64, 64], [2, 120, 64]]
    cloud_a = rng.integers(60, 90, size=(200, 3))                                 # (clear of the boxes: removing a point frees its cell)
    cloud_b = np.vstack([cloud_a[:120], rng.integers(80, 110, size=(90, 3))])
    g.add_points(world(cloud_a))
    g.update_points(world(cloud_a), world(cloud_b))                               # old \\ new removed, new \\ old added
    g.remove_points(world([[64, 64, 64]]))
    added = [added[1]]
    final_points = np.vstack([np.unique(cloud_b, axis=0), added])
    fresh = capi.Grid.from_boxes(origin, dims, res, max_dist, boxes)
    fresh.add_points(world(final_points))
    assert fresh.last_edit_cells() <= whole
    assert np.array_equal(g.d2(), fresh.d2())
    # and both equal the host builder's transform (scipy EDT, smpl_amd/scenes.py build_grid) of boxes + points-as-boxes
    host = scenes.build_grid(origin, dims, res, max_dist, list(boxes) + [(tuple(world([c])[0]), (res * 0.5,) * 3) for c in final_points])
    assert np.array_equal(g.d2(), host.d2)


def test_one_point_at_512_cube_is_a_window_not_the_grid():
    """cfg 5's grid (512^3 @ 0.01 m): one more point recomputes (2 dmax + 1)^3 cells, not 134 M; the result equals a full
    rebuild; refused inputs leave the field alone."""
    import time
    from smpl_amd import capi
    cfg = scenes.config5()
    gr = cfg.grid
    g = capi.Grid.from_boxes(gr.origin, gr.dims, gr.res, gr.max_dist, cfg.boxes)
    dmax = int(np.ceil(gr.max_dist / gr.res))
    p = np.asarray(gr.origin) + np.array([[256, 200, 300]]) * gr.res
    g.add_points(p)                                                              # (first call: allocations, kernel load)
    g.remove_points(p)
    t0 = time.perf_counter()
    g.add_points(p)
    dt = time.perf_counter() - t0
    assert g.last_edit_cells() == (2 * dmax + 1) ** 3
    assert dt < 0.02, dt                                                         # milliseconds, against ~17 ms for three full passes
    full = capi.Grid.from_boxes(gr.origin, gr.dims, gr.res, gr.max_dist, cfg.boxes)
    full.add_points(p)
    assert np.array_equal(g.d2(), full.d2())
    with pytest.raises(capi.SmplxError) as err:
        g.add_points(np.array([[np.nan, 0.0, 0.0]]))
    assert err.value.code == -1
    with pytest.raises(capi.SmplxError):
        g.add_boxes([((np.inf, 0.0, 0.0), (0.1, 0.1, 0.1))])
    assert np.array_equal(g.d2(), full.d2())


def test_an_edited_grid_asks_its_spaces_for_a_new_goal(small_cfg):
    """A space caches successor lists: after an edit of its grid GetSuccs / plan refuse (SMPLX_E_STATE) until the goal is set
    again; collision queries read the edited field at once."""
    from smpl_amd import capi
    cfg = small_cfg
    gr = cfg.grid
    grid = capi.Grid.from_boxes(gr.origin, gr.dims, gr.res, gr.max_dist, cfg.boxes)
    model = capi.Model(cfg.robot_text)
    s = capi.Space(model, grid, cfg.mprim, cfg.params, 256)
    s.set_goal_joint(cfg.goal, cfg.goal_tol)
    sid = s.set_start(cfg.start)
    assert len(s.get_succs(sid)[0]) > 0
    assert s.state_valid_batch(np.array([cfg.start]))[0][0] == 1
    # an obstacle on the arm's forearm at the start configuration
    fk = s.sphere_positions(np.array([cfg.start]))[0]
    grid.add_points(fk[-1:])
    assert s.state_valid_batch(np.array([cfg.start]))[0][0] == 0                  # the checker sees it at once
    with pytest.raises(capi.SmplxError) as err:
        s.get_succs(sid)
    assert err.value.code == -5
    with pytest.raises(capi.SmplxError):
        s.plan(5.0, 1.0, 1.0, True, True, 100, 100)
    grid.remove_points(fk[-1:])
    s.set_goal_joint(cfg.goal, cfg.goal_tol)
    sid = s.set_start(cfg.start)
    assert len(s.get_succs(sid)[0]) > 0
