"""GPU probe: the config-4 shard (128 queries) through smplx_plan_multi at several host-thread counts, with the
engine's per-slice timing split (SMPLX_DEBUG_TIMING=1).  Usage: python tools/shard_probe.py [nb] [threads,...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SMPLX_DEBUG_TIMING"] = "1"
import numpy as np
from smpl_amd import capi, scenes, shard
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
threads = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8,16").split(",")]
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 128
cfg = scenes.config2()
p = cfg.params
grid = capi.Grid(cfg.grid.origin, cfg.grid.dims, cfg.grid.res, cfg.grid.max_dist, cfg.grid.d2)
model = capi.Model(cfg.robot_text)
probe = capi.Space(model, grid, cfg.mprim, p, 256)
cs, cg = scenes.config4_candidates()
S, G = scenes.config4_queries(cs, cg, probe.state_valid_batch(cs)[0], probe.state_valid_batch(cg)[0])
S, G = S[:nq], G[:nq]
for T in threads:
    spaces = []
    for a, b in zip(S, G):
        sp = capi.Space(model, grid, cfg.mprim, p, 1024)
        sp.set_goal_joint(b, cfg.goal_tol); sp.set_start(a)
        spaces.append(sp)
    res, wall = capi.Space.plan_multi(spaces, p.eps0, p.eps_final, p.eps_delta, True, True, nb, nb, host_threads=T)
    ex = sum(r["expansions"] for r in res)
    print(f"threads {T}: wall {wall:.3f}s states/s {ex / wall:.0f} committed evals/s {sum(r['committed_succ_evals'] for r in res) / wall:.3e} "
          f"gpu evals/s {sum(r['gpu_succ_evals'] for r in res) / wall:.3e} batches {sum(r['gpu_batches'] for r in res)} "
          f"solved {sum(r['solved'] for r in res)}", flush=True)
    del spaces
