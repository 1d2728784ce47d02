#!/bin/bash
# rocprofv3 kernel traces of the K2 collision micro-benchmark (bench.py k2 leg) and of the BFS at 256^3 / 512^3,
# csv under gpurun_out/$1.  rocprofv3 gets python3 directly after "--".
set -e
OUT=gpurun_out/${1:-prof_k2}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o k2 -- python3 bench.py --no-cpu --no-planner --no-shard --overlap-streams 1 --scaling-batches= --steps 10 > "$OUT/k2.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT" -o k2fetch -- python3 bench.py --no-cpu --no-planner --no-shard --overlap-streams 1 --scaling-batches= --steps 10 > "$OUT/k2fetch.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o bfs256 -- python3 tools/bfs_time.py 256 > "$OUT/bfs256.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o bfs512 -- python3 tools/bfs_time.py 512 > "$OUT/bfs512.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT" -o bfsfetch -- python3 tools/bfs_time.py 256 > "$OUT/bfsfetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT" -o bfswrite -- python3 tools/bfs_time.py 256 > "$OUT/bfswrite.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d "$OUT" -o bfssq -- python3 tools/bfs_time.py 256 > "$OUT/bfssq.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
res = {}
for stem in ("k2", "bfs256", "bfs512"):
    f = glob.glob(os.path.join(out, "**", stem + "_kernel_trace.csv"), recursive=True)
    if not f:
        continue
    per = defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        per[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size_X"])))
    d = {}
    for k, v in per.items():
        if not k.startswith("k_"):
            continue
        big = max(g for _, g in v)
        sel = [t for t, g in v if g == big] if k == "k_state_valid" else [t for t, _ in v]
        d[k] = {"launches": len(sel), "avg_us": round(sum(sel) / len(sel) / 1e3, 3), "total_us": round(sum(sel) / 1e3, 1)}
    res[stem] = d
f = glob.glob(os.path.join(out, "**", "k2fetch_counter_collection.csv"), recursive=True)
if f:
    rows = [r for r in csv.DictReader(open(f[0])) if r["Kernel_Name"] == "k_state_valid"]
    big = max(int(r["Grid_Size"]) for r in rows)
    v = [float(r["Counter_Value"]) for r in rows if int(r["Grid_Size"]) == big and r["Counter_Name"] == "FETCH_SIZE"]
    res["k2_fetch_kb_avg"] = round(sum(v) / len(v), 1)
# BFS at 256^3: HBM traffic and SQ counters of the brick passes, per BFS run (tools/bfs_time.py runs 6 of them)
bfs = {}
for stem in ("bfsfetch", "bfswrite", "bfssq"):
    f = glob.glob(os.path.join(out, "**", stem + "_counter_collection.csv"), recursive=True)
    if not f:
        continue
    for r in csv.DictReader(open(f[0])):
        if r["Kernel_Name"].startswith("k_bfs_brick") and "seed" not in r["Kernel_Name"]:
            bfs[r["Counter_Name"]] = bfs.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
if bfs:
    runs = 6.0
    res["bfs256_brick_passes_per_run"] = {k: round(v / runs, 1) for k, v in bfs.items()}
    if "FETCH_SIZE" in bfs and "WRITE_SIZE" in bfs:
        # KB; gfx950 tallies 128-byte reads at 64 bytes (MI355X_MICROARCH.md): FETCH doubled
        res["bfs256_hbm_bytes_per_run"] = int((2 * bfs["FETCH_SIZE"] + bfs["WRITE_SIZE"]) * 1024 / runs)
json.dump(res, open(os.path.join(out, "k2_bfs_summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
