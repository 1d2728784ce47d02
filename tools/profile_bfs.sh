#!/bin/bash
# rocprofv3 of the BFS at 256^3 (or $2 = 512): per-pass kernel times, then FETCH_SIZE / WRITE_SIZE in their own passes.  csv under gpurun_out/$1
set -e
OUT=gpurun_out/${1:-prof_bfs}
N=${2:-256}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o bfstrace -- python3 tools/bfs_time.py $N > "$OUT/bfstrace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT" -o bfsfetch -- python3 tools/bfs_time.py $N > "$OUT/bfsfetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT" -o bfswrite -- python3 tools/bfs_time.py $N > "$OUT/bfswrite.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
f = glob.glob(os.path.join(out, "**", "bfstrace_kernel_trace.csv"), recursive=True)[0]
rows = sorted((r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("k_bfs")), key=lambda r: int(r["Start_Timestamp"]))
per = {}
for r in rows:
    per.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in per.items():
    print(k, "launches", len(v), "avg_us", round(sum(v) / len(v) / 1e3, 2), "total_us", round(sum(v) / 1e3, 1))
w = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if r["Kernel_Name"] == "k_bfs_brick_wave"]
last_seed = max(i for i, r in enumerate(rows) if r["Kernel_Name"] == "k_bfs_brick_seed")
ws = [r for r in rows[last_seed:] if r["Kernel_Name"] == "k_bfs_brick_wave"]
print("passes of the last BFS (us):", [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1) for r in ws])
print("idle before each of them (us):", [round((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3, 1) for a, b in zip(ws[:-1], ws[1:])])
print("first start to last end (us):", round((int(ws[-1]["End_Timestamp"]) - int(ws[0]["Start_Timestamp"])) / 1e3, 1))
for stem, name in (("bfsfetch", "FETCH_SIZE"), ("bfswrite", "WRITE_SIZE")):
    g = glob.glob(os.path.join(out, "**", stem + "_counter_collection.csv"), recursive=True)
    if not g:
        continue
    tot = {}
    for r in csv.DictReader(open(g[0])):
        if r["Counter_Name"] == name and r["Kernel_Name"].startswith("k_bfs"):
            tot[r["Kernel_Name"]] = tot.get(r["Kernel_Name"], 0.0) + float(r["Counter_Value"])
    print(name, {k: round(v / 6 / 1024, 1) for k, v in tot.items()}, "MB per BFS (counter unit: KB; 6 BFS runs)")
PY
