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
