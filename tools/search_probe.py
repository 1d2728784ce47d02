"""GPU probe of the device-resident ARA* (row N2): the cfg-2 single query and the cfg-4 shard, device search against the
host-driven loop, with the workgroup's phase clock.  Usage: python tools/search_probe.py [bound] [nq]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from smpl_amd import capi, scenes

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cfg = scenes.config2()
p = cfg.params


def ticks(sp):
    c = sp.search_counters()
    tot = sum(v for k, v in c.items() if k.startswith("t_")) or 1
    return ", ".join(f"{k[2:]} {v * 1e-5:.1f} ms ({100 * v / tot:.0f}%)" for k, v in c.items() if k.startswith("t_")) + \
        f"; grows {c['grows']}, dup pushes {c['dup_pushes']}, heap cache {c['heap_cache_entries']}, guessed rounds {c['spec_rounds']} of which used {c['spec_hits']}"


for mode in ("device", "host"):
    os.environ["SMPLX_SEARCH"] = mode
    s = capi.Space.from_config(cfg, batch_states=4096)
    s.set_goal_joint(cfg.goal, cfg.goal_tol); s.set_start(cfg.start)
    for rep in range(2):
        t0 = time.time()
        r = s.plan(p.eps0, 1.0, 1.0, True, True, 40000, 40000)
        dt = time.time() - t0
        print(f"single {mode} run {rep}: {r['expansions']} expansions in {r['seconds']:.3f}s (call {dt:.3f}s) -> {r['expansions'] / r['seconds']:.0f} states/s, "
              f"{r['committed_succ_evals'] / r['seconds']:.3e} evals/s, launches/batches {r['gpu_batches']}, solved {r['solved']} cost {r['cost']}", flush=True)
        if mode == "device":
            print("   ", ticks(s), flush=True)
    del s

grid = capi.Grid(cfg.grid.origin, cfg.grid.dims, cfg.grid.res, cfg.grid.max_dist, cfg.grid.d2)
model = capi.Model(cfg.robot_text)
probe = capi.Space(model, grid, cfg.mprim, p, 256)
cs, cg = scenes.config4_candidates()
S, G = scenes.config4_queries(cs, cg, probe.state_valid_batch(cs)[0], probe.state_valid_batch(cg)[0])
S, G = S[:nq], G[:nq]
for mode, T in (("device", 1), ("device", 1), ("host", 14)):
    os.environ["SMPLX_SEARCH"] = mode
    spaces = []
    t0 = time.time()
    for a, b in zip(S, G):
        sp = capi.Space(model, grid, cfg.mprim, p, 1024)
        sp.set_goal_joint(b, cfg.goal_tol); sp.set_start(a)
        spaces.append(sp)
    t_setup = time.time() - t0
    t0 = time.time()
    res, wall = capi.Space.plan_multi(spaces, p.eps0, p.eps_final, p.eps_delta, True, True, nb, nb, host_threads=T)
    call = time.time() - t0
    ex = sum(r["expansions"] for r in res)
    print(f"shard {mode}: {nq} queries, wall {wall:.3f}s (call {call:.3f}s, setup {t_setup:.2f}s) states/s {ex / wall:.0f} committed evals/s "
          f"{sum(r['committed_succ_evals'] for r in res) / wall:.3e} launches {res[0]['gpu_batches']} solved {sum(r['solved'] for r in res)} "
          f"cost checksum {sum(r['cost'] for r in res)}", flush=True)
    if mode == "device":
        print("    query 0:", ticks(spaces[0]), flush=True)
        print("    query 5:", ticks(spaces[5 % nq]), flush=True)
    del spaces
