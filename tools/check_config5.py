"""One-off GPU check at BASELINE config 5 size: 14-DOF dual arm, 512^3 grid @ 0.01 m (SURVEY 8d cfg 5).
Compares the engine with the oracle on a sample of expansions and a bounded ARA* run, and prints timings.
Usage (GPU box): python tools/check_config5.py [n]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle_binding import Oracle  # noqa: E402
from smpl_amd import capi, scenes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
t = time.time()
cfg = scenes.config5(n=n)
print(f"scene {n}^3 built in {time.time() - t:.1f}s, d2 max {cfg.grid.d2.max()}", flush=True)
t = time.time()
s = capi.Space.from_config(cfg, batch_states=4096)
print(f"space created in {time.time() - t:.2f}s; model trees {s.model.ntrees} nodes {s.model.nnodes} pairs {s.model.npairs}", flush=True)
t = time.time()
s.set_goal_joint(cfg.goal, cfg.goal_tol)
t_bfs = time.time() - t
print(f"GPU set_goal (BFS {s.bfs_levels()} levels) {t_bfs:.3f}s", flush=True)
t = time.time()
o = Oracle(cfg)
o.set_order(chain=True)
o.set_goal_joint(cfg.goal, cfg.goal_tol)
print(f"oracle create+BFS {time.time() - t:.1f}s", flush=True)
assert np.array_equal(o.bfs_grid(), s.bfs_grid())
print("BFS grids identical", flush=True)
lim = scenes.ARM7_LIMITS + scenes.ARM7_LIMITS
Q = np.vstack([np.array(cfg.start), scenes.random_states(lim, 63, 77)])
got = s.expand_batch(Q)
for i, q in enumerate(Q):
    e = o.eval_state(q)
    assert np.array_equal(e["flags"], got["flags"][i]), i
    v = (e["flags"] & 1) != 0
    assert np.array_equal(e["coord"][v], got["coord"][i][v]) and np.array_equal(e["h"][v], got["h"][i][v])
print("expand_batch: 64 states x", got["flags"].shape[1], "primitives identical; valid", int((got["flags"] & 1).sum()), flush=True)
o.set_order(chain=False)
assert o.set_start(cfg.start) == s.set_start(cfg.start)
nb = 3000
o.search_params(cfg.params.eps0, 1.0, 1.0, True, True, nb, nb)
ro = o.plan()
rg = s.plan(cfg.params.eps0, 1.0, 1.0, True, True, nb, nb)
assert ro["cost"] == rg["cost"] and np.array_equal(ro["expansion_log"], rg["expansion_log"])
print(json.dumps({"grid": n, "expansions": rg["expansions"], "cost": rg["cost"], "solved": rg["solved"],
                  "gpu_s": round(rg["seconds"], 3), "cpu_s": round(ro["seconds"], 3), "states": s.num_states(),
                  "gpu_bfs_s": round(t_bfs, 3), "ids_identical": True}))
