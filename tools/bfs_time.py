"""GPU probe: BFS time per goal at a BASELINE grid size (brick-major records, one wave per brick).
Usage: python tools/bfs_time.py [256|512]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from smpl_amd import capi, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = scenes.config2() if n == 256 else scenes.config5()
s = capi.Space.from_config(cfg, batch_states=256)
goals = [cfg.goal]
for k in range(4):
    g = np.array(cfg.goal); g[0] += 0.1 * (k + 1); goals.append(list(g))
s.set_goal_joint(goals[0], cfg.goal_tol)
t = time.perf_counter()
for g in goals:
    s.set_goal_joint(g, cfg.goal_tol)
dt = (time.perf_counter() - t) / len(goals)
print(f"grid {n}^3 set_goal (FK + upload + BFS) {dt * 1e3:.3f} ms per goal, passes/levels {s.bfs_levels()}")
