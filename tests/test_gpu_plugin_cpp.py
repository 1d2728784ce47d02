"""The C++ plugin mirror (include/smpl_amd/plugin.hpp) driven like smpl drives its plugins, on the GPU,
compared line by line with the oracle: once call by call (plugin_driver.cpp), once by an SBPL-shaped ARA* loop
that knows only GetSuccs / GetGoalHeuristic (sbpl_loop_driver.cpp)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


from smpl_amd.plugin_tools import build_driver, write_query  # noqa: E402


def test_cpp_plugin_driver_matches_oracle(small_cfg, tmp_path):
    from oracle_binding import Oracle
    cfg = small_cfg
    exe = build_driver("plugin_driver", tmp_path)
    nexp = 25
    write_query(cfg, tmp_path, [nexp])
    out = subprocess.run([exe, str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = out.stdout.decode().splitlines()
    o = Oracle(cfg)
    o.set_goal_joint(cfg.goal, cfg.goal_tol)
    sid = o.set_start(cfg.start)
    ev, _ = o.edge_valid(cfg.start, cfg.goal)
    assert lines[0] == f"start {sid} goal 0 valid 1 edge {int(ev)}"
    assert lines[1] == f"interp {o.waypoint_count(cfg.start, cfg.goal)}"
    frontier = [sid]
    k = 2
    for i in range(nexp):
        if i >= len(frontier):
            break
        s, c = o.get_succs(frontier[i])
        exp = f"succs {frontier[i]} :"
        for a, b in zip(s, c):
            hq = 0 if a == 0 else o.heuristic_q(o.get_state(int(a))[0])
            exp += f" {a}/{b}/{hq}"
            if a != 0 and a not in frontier:
                frontier.append(int(a))
        assert lines[k] == exp
        k += 1
    if len(frontier) > 5:
        P = np.array([o.get_state(frontier[i])[0] for i in (0, 1, 5)])
        want, _, _ = o.post_process(P, True, True, True)
        exp = f"post {len(want)}" + "".join(f" {v:.17g}" for v in want[-1]) + "".join(f" {v:.17g}" for v in want[len(want) // 2])
        assert lines[k] == exp
        k += 1
    assert lines[k] == "done"


@pytest.mark.parametrize("eps0,bound", [(5.0, (6000, 3000)), (2.0, (2500, 2500))])
def test_sbpl_shaped_arastar_loop_over_the_plugin_mirror(small_cfg, tmp_path, eps0, bound):
    """An ARA* that only knows RobotPlanningSpace::GetSuccs / GetGoalHeuristic (what smpl's unchanged ARAStar knows),
    wired like PlannerInterface wires its plugins (init, insertHeuristic, setGoal(GoalConstraint), setStart): expansion
    log, path and cost equal the oracle's; the metric distances come from the RobotHeuristic virtuals."""
    from oracle_binding import Oracle
    cfg = small_cfg
    exe = build_driver("sbpl_loop_driver", tmp_path)
    write_query(cfg, tmp_path, [eps0, 1.0, 1.0, bound[0], bound[1]])
    out = subprocess.run([exe, str(tmp_path), "log"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = {l.split(" ", 1)[0]: l.split(" ", 1)[1] if " " in l else "" for l in out.stdout.decode().splitlines()}
    o = Oracle(cfg)
    o.set_goal_joint(cfg.goal, cfg.goal_tol)
    o.set_start(cfg.start)
    o.search_params(eps0, 1.0, 1.0, True, True, bound[0], bound[1])
    e = o.plan()
    solved, cost, nexp, plen, secs, eps = lines["result"].split()
    assert int(solved) == e["ok"] and int(nexp) == e["expansions"]
    assert np.array_equal(np.array(lines["log"].split(), dtype=np.int64), e["expansion_log"])
    if e["ok"]:
        assert int(cost) == e["cost"] and float(eps) == e["eps"]
        assert np.array_equal(np.array(lines["path"].split(), dtype=np.int64), e["path"])
    # getMetricGoalDistance at the goal pose is 0 (the seeded BFS cell); getMetricStartDistance is the Manhattan
    # cell distance to the start's planning link times the resolution (bfs_heuristic.cpp:103-138)
    gp = o.goal_pose()
    mg, ms = [float(x) for x in lines["metric"].split()]
    assert mg == o.metric_goal_distance(*gp) == 0.0
    cs, cg = o.world_to_grid(*o.planning_fk(cfg.start)), o.world_to_grid(*gp)
    assert ms == cfg.grid.res * float(np.abs(cs.astype(np.int64) - cg).sum())
    assert "done" in lines
