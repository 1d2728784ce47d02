"""GPU probe, row N1: time to build a field from a box list and to edit it by one point (the window of the edit),
at a BASELINE grid size.  Usage: python tools/field_time.py [256|512|150]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from smpl_amd import capi, scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = {256: scenes.config2, 512: scenes.config5, 150: scenes.config3}[n]()
gr = cfg.grid
t0 = time.perf_counter(); g = capi.Grid.from_boxes(gr.origin, gr.dims, gr.res, gr.max_dist, cfg.boxes); t1 = time.perf_counter()
print(f"{n}^3: first build (allocations + box fill + three passes) {1e3 * (t1 - t0):.2f} ms")
g2 = capi.Grid.empty(gr.origin, gr.dims, gr.res, gr.max_dist)
t0 = time.perf_counter(); g2.add_boxes(cfg.boxes); t1 = time.perf_counter()
print(f"{n}^3: add_boxes of {len(cfg.boxes)} boxes on an existing grid {1e3 * (t1 - t0):.2f} ms, window {g2.last_edit_cells()} cells")
p = np.asarray(gr.origin) + np.array([[n // 2, n // 3, n // 2]]) * gr.res
for rep in range(3):
    t0 = time.perf_counter(); g.add_points(p); t1 = time.perf_counter(); g.remove_points(p); t2 = time.perf_counter()
    print(f"{n}^3: add one point {1e3 * (t1 - t0):.3f} ms, remove it {1e3 * (t2 - t1):.3f} ms, window {g.last_edit_cells()} cells "
          f"of {n ** 3} ({100.0 * g.last_edit_cells() / n ** 3:.2f} %)")
g.set_ref_counted(True)
t0 = time.perf_counter(); g.add_points(np.vstack([p, p, p])); t1 = time.perf_counter()
print(f"{n}^3, reference counts on: add the point three times {1e3 * (t1 - t0):.3f} ms")
