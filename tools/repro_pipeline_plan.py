import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from smpl_amd import capi, scenes
cfg=scenes.config_small()
s=capi.Space.from_config(cfg, batch_states=256, no_small_kernel=True)
s.set_goal_joint(cfg.goal,cfg.goal_tol); s.set_start(cfg.start)
r=s.plan(5,1,1,True,True,6000,3000)
print(r['expansions'], r['cost'])
