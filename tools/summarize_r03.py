"""Condenses gpurun_out/r03_prof (tools/profile_r03.sh on MI355X) into profiles/r03_*: per-kernel rocprofv3 statistics of the
bench step, the collision micro-benchmarks, the BFS at 256^3 / 512^3 and the device-resident search, plus HBM traffic from the
FETCH_SIZE / WRITE_SIZE passes (FETCH doubled: gfx950 tallies 128-byte requests at 64 bytes, MI355X_MICROARCH.md; checked here on
k_bfs_reset, whose reads are known)."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r03_prof")
DST = os.path.join(ROOT, "profiles")


def find(*parts):
    f = glob.glob(os.path.join(SRC, *parts), recursive=True)
    return f[0] if f else None


def trace_rows(path):
    rows = [r for r in csv.DictReader(open(path)) if r["Kernel_Name"].startswith("k_")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


def stats(rows):
    per = defaultdict(list)
    meta = {}
    for r in rows:
        per[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta[r["Kernel_Name"]] = {"block": int(r["Workgroup_Size_X"]), "vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]),
                                  "lds_bytes": int(r["LDS_Block_Size"]), "scratch": int(r["Scratch_Size"])}
    return {k: dict(launches=len(v), avg_us=round(sum(v) / len(v) / 1e3, 3), min_us=round(min(v) / 1e3, 3), max_us=round(max(v) / 1e3, 3),
                    total_us=round(sum(v) / 1e3, 1), **meta[k]) for k, v in per.items()}


def counter(path, name, pred):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name and pred(r):
            tot += float(r["Counter_Value"]); n += 1
    return tot, n


def write_csv(path, st):
    with open(path, "w") as f:
        f.write("name,calls,avg_us,min_us,max_us,total_us,block,vgpr,sgpr,lds_bytes,scratch\n")
        for k, v in sorted(st.items(), key=lambda kv: -kv[1]["total_us"]):
            f.write('"%s",%d,%.3f,%.3f,%.3f,%.1f,%d,%d,%d,%d,%d\n' % (k, v["launches"], v["avg_us"], v["min_us"], v["max_us"], v["total_us"],
                                                                     v["block"], v["vgpr"], v["sgpr"], v["lds_bytes"], v["scratch"]))


out = {"what": "rocprofv3 on MI355X, tools/profile_r03.sh; durations in microseconds (kernel trace End - Start), PMC in their own passes"}
# ---- the bench step (tools/profile_round.sh) ----
for name in ("summary.json", "kernel_stats.csv"):
    p = os.path.join(SRC, "round", name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(DST, "r03_" + ("round_" + name if name == "summary.json" else name)))
# ---- collision micro-benchmarks ----
k2t = find("**", "k2_kernel_trace.csv")
if k2t:
    rows = trace_rows(k2t)
    sv = [r for r in rows if r["Kernel_Name"] == "k_state_valid"]
    big = max(int(r["Grid_Size_X"]) for r in sv)
    full = [r for r in sv if int(r["Grid_Size_X"]) == big]          # 2^20 configurations: the k2 leg's launches, then the k2_pr2 leg's
    half = len(full) // 2
    d = lambda rs: [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs]
    a, b = d(full[:half]), d(full[half:])
    out["k2"] = {"kernel": "k_state_valid", "configs": big, "launches": len(a), "avg_us": round(sum(a) / len(a) / 1e3, 2), "min_us": round(min(a) / 1e3, 2),
                 "vgpr": int(full[0]["VGPR_Count"]), "lds_bytes": int(full[0]["LDS_Block_Size"]), "scene": "7-DOF arm, cfg-2 scene (table, shelf), q ~ U[limits]"}
    out["k2_pr2"] = {"kernel": "k_state_valid", "configs": big, "launches": len(b), "avg_us": round(sum(b) / len(b) / 1e3, 2), "min_us": round(min(b) / 1e3, 2),
                     "vgpr": int(full[-1]["VGPR_Count"]), "lds_bytes": int(full[-1]["LDS_Block_Size"]),
                     "scene": "PR2 right arm from data files, empty world (the shape of the reference's benchmark_cc)"}
    write_csv(os.path.join(DST, "r03_k2_kernel_stats.csv"), stats(rows))
    kf = find("**", "k2fetch_counter_collection.csv")
    if kf:
        fr = [r for r in csv.DictReader(open(kf)) if r["Kernel_Name"] == "k_state_valid" and r["Counter_Name"] == "FETCH_SIZE" and int(r["Grid_Size"]) == big]
        h = len(fr) // 2
        for key, part in (("k2", fr[:h]), ("k2_pr2", fr[h:])):
            if part:
                kb = sum(float(r["Counter_Value"]) for r in part) / len(part)
                out[key]["fetch_bytes_per_launch_x2"] = int(2 * kb * 1024)
# ---- BFS ----
for n in (256, 512):
    t = find("bfs%d" % n, "**", "bfstrace_kernel_trace.csv")
    if not t:
        continue
    rows = [r for r in trace_rows(t) if r["Kernel_Name"].startswith("k_bfs")]
    st = stats(rows)
    write_csv(os.path.join(DST, "r03_bfs%d_kernel_stats.csv" % n), st)
    seeds = [i for i, r in enumerate(rows) if r["Kernel_Name"] == "k_bfs_brick_seed"]
    last = [r for r in rows[seeds[-1]:] if r["Kernel_Name"] == "k_bfs_brick_wave"]
    runs = len(seeds)
    e = {"cells": n ** 3, "bfs_runs_profiled": runs, "passes_last_run": len(last),
         "last_run_first_start_to_last_end_us": round((int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3, 1),
         "k_bfs_brick_wave": {k: st["k_bfs_brick_wave"][k] for k in ("launches", "avg_us", "total_us", "vgpr", "lds_bytes")},
         "k_bfs_reset_launches": st.get("k_bfs_reset", {}).get("launches", 0)}
    for stem, cname in (("bfsfetch", "FETCH_SIZE"), ("bfswrite", "WRITE_SIZE")):
        c = find("bfs%d" % n, "**", stem + "_counter_collection.csv")
        if c:
            tot, _ = counter(c, cname, lambda r: r["Kernel_Name"] == "k_bfs_brick_wave")
            e[cname + "_KB_per_run"] = round(tot / runs, 1)
    if "FETCH_SIZE_KB_per_run" in e and "WRITE_SIZE_KB_per_run" in e:
        e["hbm_bytes_per_run_fetch_x2_plus_write"] = int((2 * e["FETCH_SIZE_KB_per_run"] + e["WRITE_SIZE_KB_per_run"]) * 1024)
        e["algorithmic_bytes_8_per_cell"] = 8 * n ** 3
        e["traffic_over_algorithmic"] = round(e["hbm_bytes_per_run_fetch_x2_plus_write"] / e["algorithmic_bytes_8_per_cell"], 2)
    out["bfs%d" % n] = e
# ---- device-resident search ----
s = find("**", "search_kernel_trace.csv")
if s:
    rows = trace_rows(s)
    st = stats(rows)
    write_csv(os.path.join(DST, "r03_search_kernel_stats.csv"), st)
    ks = [r for r in rows if r["Kernel_Name"] == "k_search"]
    out["search"] = {"what": "tools/search_probe.py 20000 128: the cfg-2 query alone, then the 128-query shard twice, on k_search (one workgroup per query)",
                     "k_search": st.get("k_search"), "k_search_grids": sorted(set(int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) for r in ks))}
    log = os.path.join(SRC, "search.log")
    if os.path.exists(log):
        out["search"]["probe_lines"] = [l.strip() for l in open(log) if l.startswith("shard") or l.startswith("single")][:8]
json.dump(out, open(os.path.join(DST, "r03_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
