"""The N > 1 path of bench.py on CPU: world_size 2, gloo.  Queries shard by rank with no data-path collective;
only the per-rank result records are all-gathered."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    import bench
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    # each rank owns its own query: the shard helper must give distinct, deterministic goals
    g0, g1 = bench.rank_goal_shift(0), bench.rank_goal_shift(1)
    assert g0 == [0] * 7 and g1 != g0 and all(c %% 4 == 0 for c in g1[4:])
    rec = torch.tensor([1000.0 * (rank + 1), 0.5 + 0.25 * rank, 900.0 * (rank + 1)], dtype=torch.float64)
    allrec = bench.gather_records(rec, dist, world)
    value, tmax, total = bench.aggregate(allrec)
    if rank == 0:
        print(json.dumps({"value": value, "tmax": tmax, "total": total, "shape": list(allrec.shape)}))
    dist.destroy_process_group()
""") % ROOT


def test_two_rank_gloo_aggregation(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    import json
    line = [l for l in out.stdout.decode().splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["shape"] == [2, 3]
    assert r["total"] == 3000.0 and r["tmax"] == 0.75        # whole-job units / max-over-ranks time
    assert abs(r["value"] - 4000.0) < 1e-9
