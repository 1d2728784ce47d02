"""Condenses the csv output of tools/profile_round.sh into one JSON summary + a per-kernel csv (copied to profiles/)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

STEP_KERNELS = ["k_pipe_prep", "k_pipe_setup", "k_pipe_configs", "k_pipe_finish"]


def find(out, stem, kind):
    f = glob.glob(os.path.join(out, "**", "%s_%s.csv" % (stem, kind)), recursive=True)
    return f[0] if f else None


def kernel_trace(path):
    """Durations per kernel.  The bench builds its frontier with a real search first, which launches the same kernels
    on small batches: for the kernels of the timed step only the launches with the largest grid (the B=4096 step) count."""
    rows = list(csv.DictReader(open(path)))
    big = defaultdict(int)
    for r in rows:
        big[r["Kernel_Name"]] = max(big[r["Kernel_Name"]], int(r["Grid_Size_X"]))
    per = defaultdict(list)
    meta = {}
    for r in rows:
        n = r["Kernel_Name"]
        if n in STEP_KERNELS and int(r["Grid_Size_X"]) != big[n]:
            continue
        per[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta[n] = {"grid_threads": int(r["Grid_Size_X"]), "block": int(r["Workgroup_Size_X"]), "vgpr": int(r["VGPR_Count"]),
                   "sgpr": int(r["SGPR_Count"]), "lds_bytes": int(r["LDS_Block_Size"]), "scratch": int(r["Scratch_Size"])}
    return per, meta


def counters(path):
    acc = defaultdict(lambda: defaultdict(list))
    if not path:
        return acc
    rows = list(csv.DictReader(open(path)))
    big = defaultdict(int)
    for r in rows:
        big[r["Kernel_Name"]] = max(big[r["Kernel_Name"]], int(r["Grid_Size"]))
    for r in rows:
        if int(r["Grid_Size"]) != big[r["Kernel_Name"]]:
            continue   # launches on small batches while the bench builds its frontier
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main(out, summary_path, stats_path):
    per, meta = kernel_trace(find(out, "trace", "kernel_trace"))
    rows = ["name,calls,avg_us,min_us,max_us,total_us,grid_threads,block,vgpr,sgpr,lds_bytes,scratch"]
    kernels = {}
    for n, d in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        # the timed steps are the last launches of a kernel; warm-up launches are included in the averages
        m = meta[n]
        kernels[n] = dict(launches=len(d), avg_us=round(sum(d) / len(d) / 1e3, 3), min_us=round(min(d) / 1e3, 3),
                          max_us=round(max(d) / 1e3, 3), **m)
        rows.append('"%s",%d,%.3f,%.3f,%.3f,%.1f,%d,%d,%d,%d,%d,%d' % (n[:80], len(d), sum(d) / len(d) / 1e3, min(d) / 1e3,
                                                                     max(d) / 1e3, sum(d) / 1e3, m["grid_threads"], m["block"],
                                                                     m["vgpr"], m["sgpr"], m["lds_bytes"], m["scratch"]))
    open(stats_path, "w").write("\n".join(rows) + "\n")
    pmc = {}
    for stem in ("fetch", "write", "sq1", "sq2"):
        for k, cs in counters(find(out, stem, "counter_collection")).items():
            if k in STEP_KERNELS:
                for c, v in cs.items():
                    pmc.setdefault(k, {})[c + "_avg"] = round(sum(v) / len(v), 2)
    s = {"command": "tools/profile_round.sh (rocprofv3 --kernel-trace --stats; --pmc FETCH_SIZE / WRITE_SIZE / SQ_* in "
                    "separate passes) over: python3 bench.py --no-cpu --no-planner --no-shard --no-k2 --overlap-streams 1 --scaling-batches= --steps 50",
         "timed_step_kernels": {k: kernels[k] for k in STEP_KERNELS if k in kernels},
         "other_kernels": {k: v for k, v in kernels.items() if k not in STEP_KERNELS and k.startswith("k_")},
         "pmc_per_launch": pmc}
    c = pmc.get("k_pipe_configs", {})
    if "FETCH_SIZE_avg" in c and "WRITE_SIZE_avg" in c:
        # FETCH_SIZE / WRITE_SIZE are in KB; gfx950 reports half of wide reads (MI355X_MICROARCH.md, HBM section)
        s["k_pipe_configs_traffic_bytes_per_launch"] = int((2 * c["FETCH_SIZE_avg"] + c["WRITE_SIZE_avg"]) * 1024)
        s["traffic_note"] = ("FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); the correction is "
                             "calibrated for wide coalesced reads, the 2-byte gathers here may be over-corrected: upper bound")
    try:
        s["bench_line"] = json.loads(open(os.path.join(out, "bench_line.json")).readline())
    except Exception as e:   # noqa
        s["bench_line_error"] = str(e)
    json.dump(s, open(summary_path, "w"), indent=1)
    print(json.dumps(s["timed_step_kernels"], indent=1))
    print(json.dumps(pmc, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
