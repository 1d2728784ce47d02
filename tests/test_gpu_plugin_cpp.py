"""The C++ plugin mirror (include/smpl_amd/plugin.hpp) driven like smpl drives its plugins, on the GPU,
compared line by line with the oracle."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_plugin_driver_matches_oracle(small_cfg, tmp_path):
    from oracle_binding import Oracle
    from smpl_amd import build
    cfg = small_cfg
    lib = build.build()
    exe = tmp_path / "plugin_driver"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "plugin_driver.cpp"), "-o", str(exe),
                           lib, f"-Wl,-rpath,{os.path.dirname(lib)}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib",
                           "-lamdhip64"])
    (tmp_path / "robot.txt").write_text(cfg.robot_text)
    (tmp_path / "mprim.txt").write_text(cfg.mprim)
    np.ascontiguousarray(cfg.grid.d2, np.int32).tofile(tmp_path / "grid.bin")
    p, g = cfg.params, cfg.grid
    nexp = 25
    fields = [*g.origin, *g.dims, g.res, g.max_dist, 7, *p.resolutions, p.bfs_radius, p.cost_per_cell, int(p.use_short),
              p.short_thresh, int(p.use_xyzrpy_snap), p.xyzrpy_thresh, int(p.xy_rotate_by_var3), int(p.use_long_and_short),
              *cfg.start, *cfg.goal, *cfg.goal_tol, nexp]
    (tmp_path / "query.txt").write_text(" ".join(repr(float(x)) if isinstance(x, float) else str(x) for x in fields))
    out = subprocess.run([str(exe), str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
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
