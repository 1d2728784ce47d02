"""GPU soak of the device-resident ARA*: random goals on config_small, each planned on the device kernel (as a shard of its
own: one launch for all) and by the oracle; expansion logs, costs and paths must agree.  Usage: python tools/search_soak.py [n] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from smpl_amd import capi, scenes
from oracle_binding import Oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 99
cfg = scenes.config_small()
rng = np.random.default_rng(seed)
grid = capi.Grid(cfg.grid.origin, cfg.grid.dims, cfg.grid.res, cfg.grid.max_dist, cfg.grid.d2)
model = capi.Model(cfg.robot_text)
probe = capi.Space(model, grid, cfg.mprim, cfg.params, 256)
goals = []
while len(goals) < n:
    g = np.array(cfg.start) + rng.uniform(-0.9, 0.9, size=len(cfg.start))
    if probe.state_valid_batch(g[None, :])[0][0]:
        goals.append(list(g))
spaces = []
for g in goals:
    sp = capi.Space(model, grid, cfg.mprim, cfg.params, 256)
    sp.set_goal_joint(g, cfg.goal_tol); sp.set_start(cfg.start)
    spaces.append(sp)
os.environ["SMPLX_SEARCH"] = "device"
eps = (5.0, 1.0, 1.0, True, True, 6000, 9000)
res, wall = capi.Space.plan_multi(spaces, *eps)
bad = 0
for k, (g, r) in enumerate(zip(goals, res)):
    o = Oracle(cfg)
    o.set_goal_joint(g, cfg.goal_tol); o.set_start(cfg.start)
    o.search_params(*eps)
    e = o.plan()
    same = (e["ok"] == r["solved"] and e["cost"] == r["cost"] and e["expansions"] == r["expansions"]
            and np.array_equal(e["expansion_log"], r["expansion_log"]) and np.array_equal(e["path"], r["path"]))
    bad += 0 if same else 1
    print(f"query {k}: solved {r['solved']} cost {r['cost']} expansions {r['expansions']} eps {r['satisfied_eps']} {'OK' if same else 'DIFFERS'}", flush=True)
print(f"{n - bad} of {n} identical to the oracle; device wall {wall:.3f}s")
sys.exit(1 if bad else 0)
