#!/usr/bin/env python3
"""bench.py -- successor evaluations/s of the ARA* state-expansion hot path on MI355X.

A "step" is one frontier-batched expansion: B = 4096 open states of a real ARA* search on the
config-2 scene (7-DOF arm, 256^3 voxel grid @ 0.02 m, tabletop + 64 seeded boxes), every motion
primitive applied to every state (k_state_prep + k_expand), inputs and outputs resident in HBM.
`value` = successor evaluations (state x active primitive, the loop body of
smpl/src/graph/manip_lattice.cpp:263-305) per second, whole job.

N > 1 (driver: torch.distributed.run, one rank per GPU, RCCL): the batched-query shard of
BASELINE config 4 -- every rank owns an independent query (its own goal, BFS grid, state table and
frontier) on the replicated scene, no collective on the data path; per-rank result records are
all-gathered at the end.  Weak scaling.

Extra objects on the JSON line: `roofline` (k_expand, HIP events on the launch stream inside the
timed region), `cpu_baseline` (the oracle, one host thread, rank 0 at N=1) and `planner`
(states expanded/s of a bounded ARA* query, GPU engine vs oracle, with the parity bits).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def rank_goal_shift(rank: int):
    """Goal offset (whole lattice cells) of rank r's query: independent queries on the replicated scene.
    Joints 4-6 move in multiples of 4 cells so the goal stays on the lattice the short primitives reach."""
    if rank == 0:
        return [0] * 7
    return [(-3 * rank) % 17 - 8, (2 * rank) % 9 - 4, (5 * rank) % 13 - 6, (-rank) % 7 - 3, 4 * (rank % 5 - 2),
            4 * (rank % 3 - 1), 4 * (rank % 7 - 3)]


def gather_records(rec, dist, world):
    """All-gather of the per-rank result record (RCCL on GPUs, gloo in the CPU test); the only collective."""
    import torch
    if dist is None or world == 1:
        return rec.detach().cpu().numpy()[None]
    allrec = [torch.zeros_like(rec) for _ in range(world)]
    dist.all_gather(allrec, rec)
    return torch.stack(allrec).cpu().numpy()


def aggregate(allrec):
    """Whole-job throughput: units of all ranks / max-over-ranks time."""
    total = float(allrec[:, 0].sum())
    tmax = float(allrec[:, 1].max())
    return total / tmax, tmax, total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--planner-expansions", type=int, default=40000)
    ap.add_argument("--multi-queries", type=int, default=32, help="queries interleaved on one GPU in the planner leg")
    ap.add_argument("--host-threads", type=int, default=4, help="host threads driving query slices in the planner_multi leg")
    ap.add_argument("--overlap-streams", type=int, default=4, help="independent batches in flight for the secondary figure")
    ap.add_argument("--profile-steps", type=int, default=0,
                    help="timed steps whose kernels are bracketed by HIP events (default: steps/8, at least 1); every "
                         "event costs the stream a marker, about 3 us, three of them per profiled step")
    ap.add_argument("--scaling-batches", type=str, default="16384,65536",
                    help="secondary figure: the same step at larger frontier batches (comma list, empty to skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-planner", action="store_true")
    ap.add_argument("--generic-kernels", action="store_true", help="skip the per-robot hiprtc build (A/B runs)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (the engine has no CPU path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from smpl_amd import capi, scenes

    # ---- inputs: config-2 scene; every rank its own query on the replicated scene ----
    cfg = scenes.config2(n=args.grid)
    goal = list(cfg.goal)
    if rank > 0:   # independent queries: shift the goal by whole lattice cells, keep it reachable
        goal = [g + c * scenes.DEG for g, c in zip(goal, rank_goal_shift(rank))]
    space = capi.Space.from_config(cfg, batch_states=args.batch, generic_kernels=args.generic_kernels)
    spec_ok, spec_note = space.specialized()
    ok, _ = space.state_valid_batch(np.array([goal]))
    if not ok[0]:
        goal = list(cfg.goal)
    space.set_goal_joint(goal, cfg.goal_tol)
    space.set_start(cfg.start)
    # a real frontier: run the search far enough to own >= B states, take the first B created
    p = cfg.params
    B = args.batch
    nwarm = max(1500, B // 3)
    warm = space.plan(p.eps0, p.eps_final, p.eps_delta, True, True, nwarm, nwarm)
    if space.num_states() <= B:
        raise SystemExit(f"search produced only {space.num_states()} states, need {B}")
    Q = np.stack([space.get_state(i)[0] for i in range(1, B + 1)])
    N, M = space.N, space.M

    dev = torch.device("cuda", local_rank)
    d_q = torch.from_numpy(Q).to(dev)
    d_flags = torch.zeros(B * M, dtype=torch.uint8, device=dev)
    d_coord = torch.zeros(B * M * N, dtype=torch.int32, device=dev)
    d_sq = torch.zeros(B * M * N, dtype=torch.float64, device=dev)
    d_h = torch.zeros(B * M, dtype=torch.int32, device=dev)
    d_cost = torch.zeros(B * M, dtype=torch.int32, device=dev)
    d_lk = torch.zeros(B * M, dtype=torch.int32, device=dev)
    d_work = torch.zeros(space.expand_work_bytes(B), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(space.counters_bytes(B) // 8, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream()

    def step():
        space.expand_batch_device(d_q.data_ptr(), B, d_flags.data_ptr(), d_coord.data_ptr(), d_sq.data_ptr(),
                                  d_h.data_ptr(), d_cost.data_ptr(), d_lk.data_ptr(), d_work.data_ptr(),
                                  d_cnt.data_ptr(), stream.cuda_stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    d_cnt.zero_()
    space.profile_begin(min(args.steps, args.profile_steps if args.profile_steps > 0 else max(1, args.steps // 8)))
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t1 = time.perf_counter()
    prep_ms, expand_ms, launches = space.profile_end()
    evals, valid, lookups_ref, lookups_done, configs, state_lookups = space.counters_read(d_cnt.data_ptr(), B)
    elapsed = t1 - t0

    # ---- whole-job aggregate: max time over ranks, sum of units ----
    rec = torch.tensor([float(evals), elapsed, float(valid)], dtype=torch.float64, device=dev)
    allrec = gather_records(rec, dist, world)   # RCCL all-gather of the per-rank result records
    value, tmax, total_evals = aggregate(allrec)

    # ---- roofline of the dominant kernel (k_pipe_configs: the collision check), this rank ----
    # SURVEY 8(d), collision kernel: algorithmic bytes = 4 B per distance-grid lookup + 8N B per configuration.
    # HIP events on the launch stream bracket the kernel inside the timed region (smplx_profile_*).
    L_ = max(args.steps, 1)        # the tallies cover every timed step
    P_ = max(launches, 1)          # the events cover the first --profile-steps of them
    cfg_per_launch = configs / L_
    lk_per_launch = (lookups_done + state_lookups) / L_
    alg_bytes = 4.0 * lk_per_launch + 8.0 * N * cfg_per_launch
    k_ms = prep_ms / P_            # first event interval = k_pipe_configs
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    # HBM-side bytes of the same kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
    # runs; gfx950 correction applied) -- kept in profiles/traffic.json, null when that file is absent
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            traffic = json.load(f)["k_pipe_configs"]["bytes_per_launch"] if args.batch == 4096 and args.grid == 256 else None
    except (OSError, KeyError, ValueError):
        traffic = None
    roofline = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic, "kernel": "k_pipe_configs",
                "kernel_ms": round(k_ms, 4), "finish_kernel_ms": round(expand_ms / P_, 4), "profiled_steps": int(launches),
                "algorithmic_bytes_per_launch": int(alg_bytes), "configs_per_launch": int(cfg_per_launch),
                "lookups_per_launch": int(lk_per_launch), "evals_per_launch": int(evals / L_),
                "succ_eval_bytes_per_launch": int(evals / L_ * (20 * N + 8) + 4.0 * lookups_ref / L_),
                "reference_lookups_per_launch": int(lookups_ref / L_)}

    out = {
        "metric": "successor evaluations/sec, ManipLattice GetSuccs loop body (7-DOF arm, 256^3 voxel grid)",
        "value": round(value, 1), "unit": "successor evaluations/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * tmax / args.steps, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"cfg2: 7-DOF arm7, {args.grid}^3 grid @ 0.02 m, tabletop + 64 boxes (seed 2), "
                               f"frontier batch B={B} open states x M={M} primitives, eps 5 ARA* frontier",
                   "batch_states": B, "primitives": M, "grid": args.grid, "queries": world,
                   "parallelism": f"query-shard x{world}" if world > 1 else "single query"},
        "valid_fraction": round(valid / max(evals, 1), 4),
        "kernels": ("per-robot hiprtc build, " + spec_note[:120]) if spec_ok else "generic (" + spec_note[:200] + ")",
        "roofline": roofline,
    }

    if rank == 0 and world == 1 and args.overlap_streams > 1:
        # secondary figure: the same step with several independent frontier batches in flight (what a GPU that
        # serves many queries sees); each in-flight batch has its own buffers and stream.  Not `value`.
        S_ = args.overlap_streams
        streams = [torch.cuda.Stream() for _ in range(S_)]
        sets = []
        for _ in range(S_):
            sets.append(dict(flags=torch.zeros_like(d_flags), coord=torch.zeros_like(d_coord), sq=torch.zeros_like(d_sq),
                             h=torch.zeros_like(d_h), cost=torch.zeros_like(d_cost), lk=torch.zeros_like(d_lk),
                             work=torch.zeros_like(d_work), cnt=torch.zeros_like(d_cnt)))

        def step_on(i):
            b = sets[i % S_]
            space.expand_batch_device(d_q.data_ptr(), B, b["flags"].data_ptr(), b["coord"].data_ptr(), b["sq"].data_ptr(),
                                      b["h"].data_ptr(), b["cost"].data_ptr(), b["lk"].data_ptr(), b["work"].data_ptr(),
                                      b["cnt"].data_ptr(), streams[i % S_].cuda_stream)
        for i in range(S_):
            step_on(i)
        torch.cuda.synchronize()
        nst = args.steps * S_
        t0o = time.perf_counter()
        for i in range(nst):
            step_on(i)
        torch.cuda.synchronize()
        t1o = time.perf_counter()
        out["overlapped"] = {"streams": S_, "steps": nst, "ms_per_step": round(1e3 * (t1o - t0o) / nst, 4),
                             "successor_evaluations_per_s": round(evals / max(args.steps, 1) * nst / (t1o - t0o), 1)}

    if rank == 0 and world == 1 and args.scaling_batches:
        # secondary figure: the step at larger frontier batches (the kernels are latency-bound at B=4096: about two
        # waves per SIMD).  Same scene and query; the frontier is the first B2 states a longer search creates.  Not `value`.
        out["batch_scaling"] = {}
        for B2 in [int(x) for x in args.scaling_batches.split(",") if x]:
            sp = capi.Space.from_config(cfg, batch_states=4096, generic_kernels=args.generic_kernels)
            sp.set_goal_joint(goal, cfg.goal_tol)
            sp.set_start(cfg.start)
            nw = max(1500, B2 // 3)
            while sp.num_states() <= B2 and nw <= 4 * B2:
                sp.plan(p.eps0, p.eps_final, p.eps_delta, True, True, nw, nw)
                nw *= 2
            if sp.num_states() <= B2:
                continue
            Q2 = np.stack([sp.get_state(i)[0] for i in range(1, B2 + 1)])
            t = {k: torch.zeros(B2 * M * w, dtype=dt, device=dev) for k, w, dt in
                 [("flags", 1, torch.uint8), ("coord", N, torch.int32), ("sq", N, torch.float64), ("h", 1, torch.int32),
                  ("cost", 1, torch.int32), ("lk", 1, torch.int32)]}
            q2 = torch.from_numpy(Q2).to(dev)
            w2 = torch.zeros(sp.expand_work_bytes(B2), dtype=torch.uint8, device=dev)
            c2 = torch.zeros(sp.counters_bytes(B2) // 8, dtype=torch.int64, device=dev)

            def step2():
                sp.expand_batch_device(q2.data_ptr(), B2, t["flags"].data_ptr(), t["coord"].data_ptr(), t["sq"].data_ptr(),
                                       t["h"].data_ptr(), t["cost"].data_ptr(), t["lk"].data_ptr(), w2.data_ptr(),
                                       c2.data_ptr(), stream.cuda_stream)
            for _ in range(3):
                step2()
            torch.cuda.synchronize()
            c2.zero_()
            n2 = 20
            ta = time.perf_counter()
            for _ in range(n2):
                step2()
            torch.cuda.synchronize()
            tb = time.perf_counter()
            ev2 = sp.counters_read(c2.data_ptr(), B2)[0]
            out["batch_scaling"][str(B2)] = {"ms_per_step": round(1e3 * (tb - ta) / n2, 4),
                                             "successor_evaluations_per_s": round(ev2 / (tb - ta), 1)}
            del sp, t, q2, w2, c2

    if rank == 0 and world == 1 and not args.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_binding import Oracle
        o = Oracle(cfg)
        o.set_goal_joint(goal, cfg.goal_tol)
        r = o.eval_batch_timed(Q, args.cpu_seconds)
        cpu_rate = r["evals"] / r["seconds"]
        out["cpu_baseline"] = {"value": round(cpu_rate, 1), "unit": "successor evaluations/s", "cores": 1, "kind": "port",
                               "sample": f"{r['passes']} pass(es) over the same {B}-state frontier batch "
                                         f"({r['evals']} evaluations, {r['seconds']:.1f} s), oracle/ C++ -O2, 1 thread, "
                                         f"logging/visualisation off, 4-byte cells; host has {os.cpu_count()} cores"}
        if not args.no_planner:
            nb = args.planner_expansions
            o2 = Oracle(cfg)
            o2.set_goal_joint(goal, cfg.goal_tol)
            o2.set_start(cfg.start)
            o2.search_params(p.eps0, p.eps_final, p.eps_delta, True, True, nb, nb)
            ro = o2.plan()
            sp2 = capi.Space.from_config(cfg, batch_states=args.batch, generic_kernels=args.generic_kernels)
            sp2.set_goal_joint(goal, cfg.goal_tol)
            sp2.set_start(cfg.start)
            rg = sp2.plan(p.eps0, p.eps_final, p.eps_delta, True, True, nb, nb)
            out["planner"] = {
                "query": "cfg2 single query, ARA* eps 5->1 step 1, expansion bound %d" % nb,
                "gpu_states_expanded_per_s": round(rg["expansions"] / rg["seconds"], 1),
                "cpu_states_expanded_per_s": round(ro["expansions"] / ro["seconds"], 1),
                "expansions": rg["expansions"], "path_cost": rg["cost"], "satisfied_eps": rg["satisfied_eps"],
                "gpu_seconds": round(rg["seconds"], 4), "cpu_seconds": round(ro["seconds"], 4),
                "gpu_succ_evals_total": rg["gpu_succ_evals"], "committed_succ_evals": rg["committed_succ_evals"],
                "speculative_succ_evals": rg["gpu_succ_evals"] - rg["committed_succ_evals"],
                "gpu_batches": rg["gpu_batches"], "cache_hits": rg["cache_hits"], "cache_misses": rg["cache_misses"],
                "parity": {"cost_equal": bool(ro["cost"] == rg["cost"]),
                           "expansions_equal": bool(ro["expansions"] == rg["expansions"]),
                           "expanded_ids_equal": bool(np.array_equal(ro["expansion_log"], rg["expansion_log"])),
                           "path_equal": bool(np.array_equal(ro["path"], rg["path"]))}}
            if rg["solved"]:
                # row N3: postProcessPath (interpolate -> shortcut -> interpolate) of the found path, upstream limit
                # test so that the interpolation passes do their work; GPU entry point vs the oracle's loops
                P = sp2.extract_path(rg["path"])
                sp2.post_process_path(P, True, True, True)
                t0 = time.perf_counter()
                for _ in range(5):
                    got, st = sp2.post_process_path(P, True, True, True)
                tg = (time.perf_counter() - t0) / 5
                t0 = time.perf_counter()
                want, ec, sc = o2.post_process(P, True, True, True)
                tc = time.perf_counter() - t0
                out["planner"]["post_process"] = {
                    "points_in": int(len(P)), "points_out": int(len(got)), "gpu_ms": round(tg * 1e3, 3),
                    "cpu_ms": round(tc * 1e3, 3), "gpu_configs_checked": int(st["configs"]),
                    "gpu_batches": int(st["edge_batches"]), "cpu_edge_checks": int(ec), "cpu_state_checks": int(sc),
                    "equal": bool(got.shape == want.shape and np.array_equal(got, want))}

    if rank == 0 and world == 1 and not args.no_planner and args.multi_queries > 1:
        # batched-query planner leg (BASELINE config 4 shape on one GPU): Q independent queries interleaved by one
        # host thread; aggregate committed expansions / wall time.  Parity is checked on the first query when the
        # CPU leg ran, and in tests/test_gpu_parity.py for all of them.
        nq, nb = args.multi_queries, args.planner_expansions
        spaces = []
        grid_h = capi.Grid(cfg.grid.origin, cfg.grid.dims, cfg.grid.res, cfg.grid.max_dist, cfg.grid.d2)
        model_h = capi.Model(cfg.robot_text)
        for qi in range(nq):
            g = [a + c * scenes.DEG for a, c in zip(cfg.goal, rank_goal_shift(qi))]
            sp = capi.Space(model_h, grid_h, cfg.mprim, cfg.params, args.batch,
                            generic_kernels=args.generic_kernels)   # one scene, one robot, Q queries
            okq, _ = sp.state_valid_batch(np.array([g]))
            sp.set_goal_joint(g if okq[0] else cfg.goal, cfg.goal_tol)
            sp.set_start(cfg.start)
            spaces.append(sp)
        res, wall = capi.Space.plan_multi(spaces, p.eps0, p.eps_final, p.eps_delta, True, True, nb, nb,
                                          host_threads=args.host_threads)
        tot = sum(r["expansions"] for r in res)
        out["planner_multi"] = {
            "queries": nq, "host_threads": args.host_threads, "expansion_bound_per_query": nb, "wall_seconds": round(wall, 4),
            "states_expanded_per_s": round(tot / wall, 1), "expansions_total": tot,
            "solved": int(sum(r["solved"] for r in res)),
            "gpu_succ_evals_total": int(sum(r["gpu_succ_evals"] for r in res)),
            "committed_succ_evals": int(sum(r["committed_succ_evals"] for r in res)),
            "gpu_batches": int(sum(r["gpu_batches"] for r in res)),
            "first_query_matches_single": bool("planner" in out and res[0]["cost"] == out["planner"]["path_cost"]
                                               and res[0]["expansions"] == out["planner"]["expansions"])}
        del spaces

    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
