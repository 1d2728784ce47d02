"""Query sharding for the batched-query configuration (BASELINE config 4).

The path shards by QUERY: query i of the seeded list goes to rank i // per_rank (128 per GPU), the scene (voxel grid,
robot, primitives) is replicated read-only, every query owns its goal, BFS grid, state table and OPEN list.  There is
no collective on the data path; the only exchange is one all-gather of the per-query result records at the end
(RCCL over xGMI on GPUs -- 128 records x 6 int64 per rank, latency-bound, any algorithm does; gloo in the CPU test).

This module holds the host-side pieces both bench.py and tests/test_distributed_gloo.py run: which queries a rank
owns, the record layout, the all-gather and the whole-job aggregation.
"""
from __future__ import annotations

import numpy as np

from . import scenes

# one record per query, int64: what PlannerInterface reports per solve (smpl_ros/src/ros/planner_interface.cpp:1438-1446)
REC_FIELDS = ("query", "status", "cost", "expansions", "path_len", "succ_evals")
REC_WIDTH = len(REC_FIELDS)


def rank_queries(starts, goals, rank: int, world: int, per_rank: int = 128):
    """(first, last, starts[first:last], goals[first:last]) of this rank; ranks beyond the list own nothing."""
    total = int(starts.shape[0])
    first, last = scenes.shard_range(rank, world, total, per_rank)
    first = min(first, total)
    return first, last, starts[first:last], goals[first:last]


def pack_records(first: int, results) -> np.ndarray:
    """results: one dict per owned query with solved/cost/expansions/path (or path_len)/committed_succ_evals."""
    rec = np.zeros((len(results), REC_WIDTH), np.int64)
    for k, r in enumerate(results):
        plen = r["path_len"] if "path_len" in r else len(r["path"])
        rec[k] = (first + k, int(r["solved"]), int(r["cost"]), int(r["expansions"]), int(plen),
                  int(r.get("committed_succ_evals", r.get("succ_evals", 0))))
    return rec


def gather_query_records(rec: np.ndarray, per_rank: int, dist, world: int, device=None) -> np.ndarray:
    """All-gather of the per-query records.  Every rank contributes a fixed [per_rank, REC_WIDTH] block (rows beyond
    its own queries carry query = -1) so that one all_gather does it; returns the valid rows in query order."""
    import torch
    block = np.full((per_rank, REC_WIDTH), -1, np.int64)
    block[:rec.shape[0]] = rec
    t = torch.from_numpy(block)
    if device is not None:
        t = t.to(device)
    if dist is None or world == 1:
        allrec = block[None]
    else:
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        allrec = torch.stack(out).cpu().numpy()
    rows = allrec.reshape(-1, REC_WIDTH)
    rows = rows[rows[:, 0] >= 0]
    return rows[np.argsort(rows[:, 0], kind="stable")]


def gather_scalars(values, dist, world: int, device=None) -> np.ndarray:
    """All-gather of a few float64 per rank (units processed, seconds): [world, len(values)]."""
    import torch
    t = torch.tensor([float(v) for v in values], dtype=torch.float64)
    if device is not None:
        t = t.to(device)
    if dist is None or world == 1:
        return t.cpu().numpy()[None]
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return torch.stack(out).cpu().numpy()


def aggregate(units_and_seconds: np.ndarray):
    """Whole-job throughput of a sharded run: units of all ranks / max-over-ranks time.  Input rows: (units, seconds)."""
    total = float(units_and_seconds[:, 0].sum())
    tmax = float(units_and_seconds[:, 1].max())
    return (total / tmax if tmax > 0 else 0.0), tmax, total


def summarize(rows: np.ndarray) -> dict:
    """Job-level summary of the gathered per-query records."""
    f = {n: i for i, n in enumerate(REC_FIELDS)}
    return {"queries": int(rows.shape[0]), "solved": int(rows[:, f["status"]].sum()),
            "expansions_total": int(rows[:, f["expansions"]].sum()), "succ_evals_total": int(rows[:, f["succ_evals"]].sum()),
            "cost_checksum": int(rows[:, f["cost"]].sum() % 1000000007),
            "path_len_total": int(rows[:, f["path_len"]].sum())}
