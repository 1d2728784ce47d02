import sys, os, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from smpl_amd import capi, scenes
cfg=scenes.config2()
for nosmall in (False, True, False, True):
    s=capi.Space.from_config(cfg, batch_states=4096, no_small_kernel=nosmall)
    s.set_goal_joint(cfg.goal,cfg.goal_tol); s.set_start(cfg.start)
    r=s.plan(5,1,1,True,True,40000,40000)
    print("no_small" if nosmall else "small", r['expansions'], round(r['seconds'],4), r['gpu_batches'], r['cache_misses'], r['gpu_succ_evals'], file=sys.stderr)
