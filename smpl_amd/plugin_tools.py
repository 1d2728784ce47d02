"""Helpers around the C++ plugin mirror (include/smpl_amd/plugin.hpp): build one of the small C++ drivers under
tests/cpp against the in-tree library, and write the plain-text query files they read.  Used by the GPU tests and by
bench.py's planner_plain leg."""
from __future__ import annotations

import os
import subprocess

import numpy as np

from . import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_driver(name: str, out_dir) -> str:
    lib = build.build()
    exe = os.path.join(str(out_dir), name)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", exe,
                           lib, f"-Wl,-rpath,{os.path.dirname(lib)}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib",
                           "-lamdhip64"])
    return exe


def write_query(cfg, out_dir, tail) -> None:
    """robot.txt, mprim.txt, grid.bin and query.txt (scene + params + start + goal + tolerances + `tail`)."""
    out_dir = str(out_dir)
    open(os.path.join(out_dir, "robot.txt"), "w").write(cfg.robot_text)
    open(os.path.join(out_dir, "mprim.txt"), "w").write(cfg.mprim)
    np.ascontiguousarray(cfg.grid.d2, np.int32).tofile(os.path.join(out_dir, "grid.bin"))
    p, g = cfg.params, cfg.grid
    fields = [*g.origin, *g.dims, g.res, g.max_dist, len(cfg.start), *p.resolutions, p.bfs_radius, p.cost_per_cell,
              int(p.use_short), p.short_thresh, int(p.use_xyzrpy_snap), p.xyzrpy_thresh, int(p.xy_rotate_by_var3),
              int(p.use_long_and_short), *[float(x) for x in cfg.start], *[float(x) for x in cfg.goal],
              *[float(x) for x in cfg.goal_tol], *tail]
    open(os.path.join(out_dir, "query.txt"), "w").write(
        " ".join(repr(float(x)) if isinstance(x, float) else str(x) for x in fields))
