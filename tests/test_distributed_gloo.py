"""The N > 1 path on CPU: world_size 2, gloo.  Queries shard by rank (query i -> rank i // per_rank) with no data-path
collective; only the per-query result records are all-gathered.  The ranks here run the shard
code of bench.py -- smpl_amd/shard.py: which queries a rank owns, record layout, all-gather, whole-job aggregation --
with the ORACLE standing in for the GPU planner (the product has no CPU path; the oracle is the checker and this is a
test).  The gathered records must equal a single-process run over the whole list."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import numpy as np
    import torch, torch.distributed as dist
    from smpl_amd import scenes, shard
    from oracle_binding import Oracle
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    cfg = scenes.config_small()
    # a 4-query list (seeded): 2 per rank
    S = np.load(%(queries)r)["S"]; G = np.load(%(queries)r)["G"]
    per_rank = 2
    first, last, s_mine, g_mine = shard.rank_queries(S, G, rank, world, per_rank)
    assert (first, last) == (2 * rank, 2 * rank + 2)
    results, units, secs = [], 0, 0.0
    for a, b in zip(s_mine, g_mine):
        o = Oracle(cfg)
        o.set_goal_joint(b, cfg.goal_tol); o.set_start(a)
        o.search_params(5.0, 1.0, 1.0, True, True, 800, 800)
        r = o.plan()
        results.append({"solved": r["ok"], "cost": r["cost"], "expansions": r["expansions"], "path": r["path"],
                        "succ_evals": r["succ_evals"]})
        units += r["succ_evals"]; secs += r["seconds"]
    rec = shard.pack_records(first, results)
    rows = shard.gather_query_records(rec, per_rank, dist, world)          # the one collective of the path
    sc = shard.gather_scalars([units, secs + 0.25 * rank], dist, world)
    value, tmax, total = shard.aggregate(sc)
    if rank == 0:
        print(json.dumps({"rows": rows.tolist(), "value": value, "tmax": tmax, "total": total, "sc": sc.tolist(),
                          "summary": shard.summarize(rows)}))
    dist.destroy_process_group()
""")


def test_two_rank_gloo_query_shard(tmp_path, small_cfg):
    from oracle_binding import Oracle
    from smpl_amd import scenes, shard
    cfg = small_cfg
    cells = [[-49, 7, 21, -14, -8, -12, 16], [-21, 7, 14, -7, 8, -4, 12], [-35, 14, 7, -14, 4, -8, 8], [-42, 10, 14, -10, 0, -8, 12]]
    G = np.array([[cfg.start[i] + c * scenes.DEG for i, c in enumerate(cs)] for cs in cells])
    S = np.tile(np.array(cfg.start), (4, 1))
    S[1, 0] -= 3 * scenes.DEG
    S[3, 1] += 2 * scenes.DEG
    qfile = str(tmp_path / "queries.npz")
    np.savez(qfile, S=S, G=G)
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "queries": qfile})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    line = [l for l in out.stdout.decode().splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    rows = np.array(r["rows"], dtype=np.int64)
    # single-process reference over the whole list: same records, in query order
    want = []
    for a, b in zip(S, G):
        o = Oracle(cfg)
        o.set_goal_joint(b, cfg.goal_tol); o.set_start(a)
        o.search_params(5.0, 1.0, 1.0, True, True, 800, 800)
        e = o.plan()
        want.append({"solved": e["ok"], "cost": e["cost"], "expansions": e["expansions"], "path": e["path"],
                     "succ_evals": e["succ_evals"]})
    exp = shard.pack_records(0, want)
    assert rows.shape == (4, shard.REC_WIDTH) and np.array_equal(rows, exp)
    assert list(rows[:, 0]) == [0, 1, 2, 3]
    # whole-job value = units of all ranks / max-over-ranks time
    sc = np.array(r["sc"])
    assert sc.shape == (2, 2) and r["total"] == float(exp[:, 5].sum())
    assert r["tmax"] == sc[:, 1].max() and abs(r["value"] - r["total"] / r["tmax"]) < 1e-9
    assert r["summary"]["queries"] == 4 and r["summary"]["expansions_total"] == int(exp[:, 3].sum())


def test_shard_ownership_covers_the_list_once():
    from smpl_amd import shard
    S = np.arange(1024 * 7, dtype=np.float64).reshape(1024, 7)
    seen = []
    for rank in range(8):
        first, last, s, g = shard.rank_queries(S, S, rank, 8)
        assert s.shape[0] == 128 and np.array_equal(s, S[first:last])
        seen += list(range(first, last))
    assert seen == list(range(1024))
    # fewer ranks than the list needs: a prefix (weak scaling, per-GPU work fixed); a rank beyond the list owns nothing
    assert shard.rank_queries(S, S, 1, 2)[:2] == (128, 256)
    assert shard.rank_queries(S[:130], S[:130], 1, 2)[:2] == (128, 130)
    assert shard.rank_queries(S[:100], S[:100], 1, 2)[2].shape[0] == 0
    # the fixed-size gather block tolerates a short last rank
    rec = shard.pack_records(128, [{"solved": 1, "cost": 5, "expansions": 7, "path": [1, 0], "succ_evals": 9}])
    rows = shard.gather_query_records(rec, 128, None, 1)
    assert rows.tolist() == [[128, 1, 5, 7, 2, 9]]
