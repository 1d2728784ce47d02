"""Times BFS (set_goal) on the cfg2 scene; under rocprofv3 the per-level kernel durations show where the time goes."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from smpl_amd import capi, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = scenes.config2(n=n)
s = capi.Space.from_config(cfg, generic_kernels=True)
for rep in range(3):
    t = time.time()
    s.set_goal_joint(cfg.goal, cfg.goal_tol)
    print("set_goal %.2f ms, levels %d" % ((time.time() - t) * 1e3, s.bfs_levels()), file=sys.stderr)
