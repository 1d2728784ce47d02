"""GPU probe: distance-field construction time (row N1) at the BASELINE grid sizes vs the host builder (scipy EDT)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from smpl_amd import capi, scenes
for name, mk in (("cfg2 256^3", scenes.config2), ("cfg3 150^3 (cap 90 cells)", scenes.config3), ("cfg5 512^3", scenes.config5)):
    t = time.perf_counter(); cfg = mk(); t_host = time.perf_counter() - t
    gr = cfg.grid
    g = capi.Grid.empty(gr.origin, gr.dims, gr.res, gr.max_dist)
    t = time.perf_counter(); g.add_boxes(cfg.boxes); t_gpu = time.perf_counter() - t
    t = time.perf_counter(); g.add_points(np.array([[0.3, 0.2, 1.0]])); t_pt = time.perf_counter() - t
    print(f"{name}: GPU add_boxes (fill + 3 passes + sync) {t_gpu * 1e3:.2f} ms, one more point {t_pt * 1e3:.2f} ms; host builder (scene incl. scipy EDT) {t_host:.2f} s", flush=True)
    del g
