"""GPU probe: single-query planner rate on config 2 (smplx_plan, 40 000 expansions), a few repetitions."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smpl_amd import capi, scenes
cfg = scenes.config2(); p = cfg.params
grid = capi.Grid(cfg.grid.origin, cfg.grid.dims, cfg.grid.res, cfg.grid.max_dist, cfg.grid.d2)
model = capi.Model(cfg.robot_text)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
for rep in range(3):
    s = capi.Space(model, grid, cfg.mprim, p, 4096)
    s.set_goal_joint(cfg.goal, cfg.goal_tol); s.set_start(cfg.start)
    r = s.plan(p.eps0, p.eps_final, p.eps_delta, True, True, nb, nb)
    print(f"hint_scan={os.environ.get('SMPLX_HINT_SCAN','default')}: {r['expansions'] / r['seconds']:.0f} states/s, {r['seconds']:.3f}s, misses {r['cache_misses']}, gpu evals {r['gpu_succ_evals']}, log checksum {int(r['expansion_log'].sum())}", flush=True)
