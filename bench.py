#!/usr/bin/env python3
"""bench.py -- the ARA* state-expansion hot path of smpl on MI355X: successor evaluations/s and states expanded/s.

`value` (BASELINE.json metric, config 2): one STEP = one frontier-batched expansion, B = 4096 open states of a real
ARA* search on the config-2 scene (7-DOF arm, 256^3 voxel grid @ 0.02 m, tabletop + 64 seeded boxes), every motion
primitive applied to every state -- the loop body of ManipLattice::GetSuccs (smpl/src/graph/manip_lattice.cpp:263-305)
for every (state, primitive), including the state-table lookup of every valid successor and the ballot compaction
of the valid ones (K5) -- inputs and outputs resident in HBM.  value = successor evaluations / s, whole job.

That figure is a kernel-level rate.  What a caller of the plugin API gets is reported beside it, in the same line:
  planner        one query through smplx_plan (the engine's own ARA*, frontier hints from its OPEN list)
  planner_plain  the same query driven by an SBPL-shaped ARA* that only knows GetSuccs / GetGoalHeuristic
                 (tests/cpp/sbpl_loop_driver.cpp over include/smpl_amd/plugin.hpp; no hints)
  shard          BASELINE config 4's per-GPU shard: 128 independent queries of the seeded list through
                 smplx_plan_multi -- states expanded/s and successor evaluations/s (committed and total)
  cpu_baseline   the oracle (CPU restatement) on one host core over the step's batch; cpu_shard: the same queries as
                 `shard`, one per host thread, on the oracle
  k2             the collision micro-benchmark SURVEY 8(d) states the 60 % roofline target on: 2^20 configurations
                 q ~ U[limits], std::mt19937_64 seed 12345 (benchmark_cc.cpp:280-301), through the state-validity kernel

N > 1 (driver: torch.distributed.run, one rank per GPU, RCCL): queries shard by rank -- rank r owns queries
[128 r, 128 r + 128) of the seeded config-4 list on the replicated scene (its own goals, BFS grids, state tables), no
collective on the data path; the per-query result records are all-gathered once at the end (smpl_amd/shard.py).  Every
rank times the step on a frontier of its own first query and runs its shard; `value` = step evaluations of all ranks /
max-over-ranks time (weak scaling), `shard` = whole-job figures of the config-4 run.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def host_threads_available(cap: int) -> int:
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(cap, n))


def frontier_of(space, B, p, nwarm0):
    """The first B states a bounded search creates (a real frontier, not random states)."""
    nw = nwarm0
    while space.num_states() <= B and nw <= 16 * max(B, nwarm0):
        space.plan(p.eps0, p.eps_final, p.eps_delta, True, True, nw, nw)
        nw *= 2
    if space.num_states() <= B:
        return None
    return np.stack([space.get_state(i)[0] for i in range(1, B + 1)])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="host threads of the cpu_shard leg; 0 = two runs, at nproc // 8 (a GPU's share of an 8-GPU node) and at nproc")
    ap.add_argument("--planner-expansions", type=int, default=1000000,
                    help="expansion bound of the single-query leg (the cfg-2 query reaches its goal after 194 806 expansions)")
    ap.add_argument("--queries-per-gpu", type=int, default=128, help="config-4 shard size per rank")
    ap.add_argument("--shard-expansions", type=int, default=20000, help="expansion bound per query in the shard leg")
    ap.add_argument("--host-threads", type=int, default=14,
                    help="worker threads of the HOST-DRIVEN shard leg (searches and commits; one more thread submits to the GPU); "
                         "the device-resident search needs one host thread")
    ap.add_argument("--overlap-streams", type=int, default=4, help="independent batches in flight for the secondary figure")
    ap.add_argument("--profile-steps", type=int, default=0,
                    help="timed steps whose kernels are bracketed by HIP events (default: steps/8, at least 1); every "
                         "event costs the stream a marker, about 3 us, three of them per profiled step")
    ap.add_argument("--scaling-batches", type=str, default="16384,65536",
                    help="secondary figure: the same step at larger frontier batches (comma list, empty to skip)")
    ap.add_argument("--k2-states", type=int, default=1 << 20)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-planner", action="store_true")
    ap.add_argument("--no-shard", action="store_true")
    ap.add_argument("--no-k2", action="store_true")
    ap.add_argument("--setup-threads", type=int, default=8, help="host threads that set goals (BFS) and starts of the shard's queries")
    ap.add_argument("--generic-kernels", action="store_true", help="skip the per-robot hiprtc build (A/B runs)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (the engine has no CPU path)")
    # one rank per GPU over RCCL.  Rehearsal on a box with fewer GPUs than ranks (SMPLX_BENCH_BACKEND=gloo): the ranks share
    # the devices there are and the result records travel over gloo -- same control flow, not a measurement.
    backend = os.environ.get("SMPLX_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", dev_index)
    coll_dev = dev if backend == "nccl" else None     # where the collectives' tensors live

    from smpl_amd import capi, scenes, shard

    # ---- the replicated scene and this rank's queries ---------------------------------------------------------------
    cfg = scenes.config2(n=args.grid)
    p = cfg.params
    grid_h = capi.Grid(cfg.grid.origin, cfg.grid.dims, cfg.grid.res, cfg.grid.max_dist, cfg.grid.d2)
    model_h = capi.Model(cfg.robot_text)

    def new_space(batch=args.batch):
        return capi.Space(model_h, grid_h, cfg.mprim, p, batch, generic_kernels=args.generic_kernels)

    space = new_space()
    spec_ok, spec_note = space.specialized()
    # BASELINE config 4: the seeded list of (start, goal) pairs; validity of the candidates decided by the engine
    cs, cg = scenes.config4_candidates()
    S_all, G_all = scenes.config4_queries(cs, cg, space.state_valid_batch(cs)[0], space.state_valid_batch(cg)[0])
    first, last, S_mine, G_mine = shard.rank_queries(S_all, G_all, rank, world, args.queries_per_gpu)

    # ---- the step: a frontier batch of this rank's own query (rank 0: the config-2 query itself) -------------------
    B = args.batch
    q_start, q_goal = (list(cfg.start), list(cfg.goal)) if rank == 0 or len(S_mine) == 0 else (S_mine[0], G_mine[0])
    space.set_goal_joint(q_goal, cfg.goal_tol)
    space.set_start(q_start)
    Q = frontier_of(space, B, p, max(1500, B // 3))
    if Q is None:   # that query ends before it owns B states: take the config-2 query's frontier
        q_start, q_goal = list(cfg.start), list(cfg.goal)
        space.set_goal_joint(q_goal, cfg.goal_tol)
        space.set_start(q_start)
        Q = frontier_of(space, B, p, max(1500, B // 3))
    if Q is None:
        raise SystemExit(f"search produced only {space.num_states()} states, need {B}")
    N, M = space.N, space.M

    d_q = torch.from_numpy(Q).to(dev)
    d_flags = torch.zeros(B * M, dtype=torch.uint8, device=dev)
    d_coord = torch.zeros(B * M * N, dtype=torch.int32, device=dev)
    d_sq = torch.zeros(B * M * N, dtype=torch.float64, device=dev)
    d_h = torch.zeros(B * M, dtype=torch.int32, device=dev)
    d_cost = torch.zeros(B * M, dtype=torch.int32, device=dev)
    d_lk = torch.zeros(B * M, dtype=torch.int32, device=dev)
    d_work = torch.zeros(space.expand_work_bytes(B), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(space.counters_bytes(B) // 8, dtype=torch.int64, device=dev)
    # K5 outputs: ids from the device copy of the state table (it holds the states of the search that produced the
    # frontier) and the ballot-compacted successor stream
    rb = space.compact_rec_b_bytes()
    d_id = torch.zeros(B * M, dtype=torch.int32, device=dev)
    cap_k5 = capi.lib().smplx_compact_capacity(space.h, B)
    d_reca = torch.zeros(cap_k5 * 2, dtype=torch.int32, device=dev)
    d_recb = torch.zeros(cap_k5 * rb, dtype=torch.uint8, device=dev)
    d_btab = torch.zeros(space.compact_blocks(B) * 4, dtype=torch.int32, device=dev)
    d_tot = torch.zeros(capi.lib().smplx_compact_totals_len(), dtype=torch.int32, device=dev)
    space.table_sync()
    stream = torch.cuda.current_stream()

    def step():
        space.expand_batch_k5_device(d_q.data_ptr(), B, d_flags.data_ptr(), d_coord.data_ptr(), d_sq.data_ptr(),
                                     d_h.data_ptr(), d_cost.data_ptr(), d_lk.data_ptr(), d_id.data_ptr(), d_reca.data_ptr(), cap_k5,
                                     d_recb.data_ptr(), cap_k5, d_btab.data_ptr(), d_tot.data_ptr(), d_work.data_ptr(),
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
    raw_k5 = d_tot.cpu().numpy()
    tot_k5 = [int(raw_k5[0:-1:32].sum()), int(raw_k5[1:-1:32].sum()), int(raw_k5[-1])]
    known = int((d_id >= 0).sum().item())
    k5 = {"valid_successors_per_launch": int(tot_k5[0]), "full_records_per_launch": int(tot_k5[1]), "overflow": int(tot_k5[2]),
          "known_ids_per_launch": known, "table_states": space.num_states() - 1,
          "compact_bytes_per_launch": int(8 * tot_k5[0] + rb * tot_k5[1]), "dense_bytes_per_launch": int(B * M * (12 * N + 9))}

    # whole-job aggregate: units of all ranks / max-over-ranks time (one all-gather of three doubles per rank)
    sc = shard.gather_scalars([float(evals), elapsed, float(valid)], dist, world, coll_dev)
    value, tmax, total_evals = shard.aggregate(sc[:, :2])

    # ---- roofline of the dominant kernel (k_pipe_configs: the collision check), this rank ---------------------------
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
                               f"frontier batch B={B} open states x M={M} primitives, eps 5 ARA* frontier; fork "
                               f"semantics (xy rotation by state[3])",
                   "batch_states": B, "primitives": M, "grid": args.grid, "queries": world,
                   "parallelism": f"query-shard x{world} (cfg 4 list, 128 queries per GPU)" if world > 1 else "single query"},
        "value_is": "kernel-level rate of the batched step (inputs resident in HBM); rates through the plugin API are in "
                    "planner / planner_plain / shard",
        "valid_fraction": round(valid / max(evals, 1), 4),
        "k5": k5,
        "kernels": ("per-robot hiprtc build, " + spec_note[:120]) if spec_ok else "generic (" + spec_note[:200] + ")",
        "roofline": roofline,
    }
    single = rank == 0 and world == 1

    if single and args.overlap_streams > 1:
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
        del sets, streams

    if single and args.scaling_batches:
        # secondary figure: the step at larger frontier batches (the kernels are latency-bound at B=4096: about two
        # waves per SIMD).  Same scene and query; the frontier is the first B2 states a longer search creates.  Not `value`.
        out["batch_scaling"] = {}
        for B2 in [int(x) for x in args.scaling_batches.split(",") if x]:
            sp = new_space(4096)
            sp.set_goal_joint(q_goal, cfg.goal_tol)
            sp.set_start(q_start)
            Q2 = frontier_of(sp, B2, p, max(1500, B2 // 3))
            if Q2 is None:
                continue
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

    Oracle = None
    if single and not args.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_binding import Oracle   # the checker; here only as the timed CPU baseline and the parity bits

    if single and not args.no_k2:
        # ---- K2: the collision micro-benchmark: 2^20 random configurations through the validity kernel ----
        n2 = args.k2_states
        Qk = scenes.benchmark_states(scenes.ARM7_LIMITS, n2, 12345)
        dq = torch.from_numpy(Qk).to(dev)
        dv = torch.zeros(n2, dtype=torch.uint8, device=dev)
        dl = torch.zeros(n2, dtype=torch.int32, device=dev)
        for _ in range(2):
            space.state_valid_batch_device(dq.data_ptr(), n2, dv.data_ptr(), dl.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        reps = 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            space.state_valid_batch_device(dq.data_ptr(), n2, dv.data_ptr(), dl.data_ptr(), stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        lk_total = int(dl.sum(dtype=torch.int64).item())
        nvalid = int(dv.sum(dtype=torch.int64).item())
        k2_bytes = 4.0 * lk_total + 8.0 * N * n2
        k2_traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                k2_traffic = json.load(f)["k_state_valid"]["bytes_per_launch"] if n2 == (1 << 20) and args.grid == 256 else None
        except (OSError, KeyError, ValueError):
            k2_traffic = None
        k2 = {"kernel": "k_state_valid", "configs": n2, "inputs": "q ~ U[limits]: same distribution as benchmark_cc.cpp:280-301, SURVEY-specified engine and seed (std::mt19937_64, 12345); parity against the oracle only",
              "kernel_ms": round(ms, 4), "collision_checks_per_s": round(n2 / (ms * 1e-3), 1), "valid_fraction": round(nvalid / n2, 4),
              "lookups": lk_total, "algorithmic_bytes": int(k2_bytes),
              "roofline": {"bound": "hbm", "achieved": round(k2_bytes / (ms * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": round(k2_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6), "traffic": k2_traffic}}
        if Oracle is not None:
            o = Oracle(cfg)
            o.set_order(chain=True)
            ns = min(n2, 200000)
            ok_c, lk_c, sec = o.state_valid_batch_timed(Qk[:ns])
            k2["cpu_checks_per_s_one_core"] = round(ns / sec, 1)
            k2["cpu_sample"] = f"first {ns} of the same states, oracle, 1 thread ({sec:.1f} s)"
            k2["parity"] = {"valid_bits_equal": bool(np.array_equal(ok_c, dv[:ns].cpu().numpy())),
                            "lookups_equal_on_valid": bool(np.array_equal(lk_c[ok_c == 1], dl[:ns].cpu().numpy()[ok_c == 1]))}
            del o
        out["k2"] = k2
        del dq, dv, dl

        # ---- a second timed frontier whose edges DO collide: B valid states drawn uniformly within the limits (the K2
        # generator, other seed) -- arms folded next to the table and the shelf, so that colliding edges, joint limits and the
        # sparse side of the compaction are inside a timed step (the search frontier above is almost collision-free) ----
        Qr_all = scenes.benchmark_states(scenes.ARM7_LIMITS, 4 * B, 777)
        ok_r, _ = space.state_valid_batch(Qr_all)
        Qr = np.ascontiguousarray(Qr_all[ok_r.astype(bool)][:B])
        if Qr.shape[0] == B:
            d_qr = torch.from_numpy(Qr).to(dev)
            cnt_r = torch.zeros_like(d_cnt)

            def step_random():
                space.expand_batch_k5_device(d_qr.data_ptr(), B, d_flags.data_ptr(), d_coord.data_ptr(), d_sq.data_ptr(),
                                             d_h.data_ptr(), d_cost.data_ptr(), d_lk.data_ptr(), d_id.data_ptr(), d_reca.data_ptr(), cap_k5,
                                             d_recb.data_ptr(), cap_k5, d_btab.data_ptr(), d_tot.data_ptr(), d_work.data_ptr(),
                                             cnt_r.data_ptr(), stream.cuda_stream)
            for _ in range(5):
                step_random()
            torch.cuda.synchronize()
            cnt_r.zero_()
            nr = max(args.steps, 200)
            tr0 = time.perf_counter()
            for _ in range(nr):
                step_random()
            torch.cuda.synchronize()
            tr1 = time.perf_counter()
            ev_r, va_r, lkr_r, lkd_r, cf_r, sl_r = space.counters_read(cnt_r.data_ptr(), B)
            flr = d_flags.cpu().numpy().reshape(B, M)
            rf = {"states": f"{B} valid states q ~ U[limits] (K2 generator, seed 777)", "steps": nr,
                  "ms_per_step": round(1e3 * (tr1 - tr0) / nr, 4), "successor_evaluations_per_s": round(ev_r / (tr1 - tr0), 1),
                  "valid_fraction": round(va_r / max(ev_r, 1), 4),
                  "edges_out_of_limits": int(((flr & 0x20) != 0).sum()), "edges_in_collision": int(((flr & 0x40) != 0).sum()),
                  "edges_inactive": int(((flr & 0x10) != 0).sum()), "configs_per_launch": int(cf_r / nr)}
            if Oracle is not None:
                o = Oracle(cfg)
                o.set_order(chain=True)
                o.set_goal_joint(q_goal, cfg.goal_tol)
                hc = d_h.cpu().numpy().reshape(B, M); cc = d_cost.cpu().numpy().reshape(B, M)
                same = True
                for i in range(64):
                    e = o.eval_state(Qr[i])
                    v = (e["flags"] & 1) != 0
                    same = same and bool(np.array_equal(e["flags"], flr[i]) and np.array_equal(e["h"][v], hc[i][v]) and np.array_equal(e["cost"][v], cc[i][v]))
                rf["parity"] = {"flags_h_cost_equal_on_first_64_states": same}
                del o
            out["random_frontier"] = rf
            del d_qr, cnt_r

        # ---- the reference's own collision benchmark shape ("collision checks / s, PR2 right arm, uniformly
        # random joint states", benchmark_cc.cpp:234-301): the PR2 right arm built from data files (the reference's
        # collision_model_pr2.yaml + a URDF subset, tests/golden/) in an EMPTY world ----
        gold = os.path.join(ROOT, "tests", "golden")
        try:
            pr2 = scenes.pr2_right_arm_text(open(os.path.join(gold, "collision_model_pr2.yaml")).read(),
                                            open(os.path.join(gold, "pr2_right_arm.urdf")).read(),
                                            json.load(open(os.path.join(gold, "pr2_right_arm_acm.json")))["allowed_pairs"])
        except OSError:
            pr2 = None
        if pr2 is not None:
            import dataclasses
            lim = [(-2.1353981634, 0.564601836603), (-0.3536, 1.2963), (-3.75, 0.65), (-2.1213, -0.15), (-np.pi, np.pi), (-2.0, -0.1),
                   (-np.pi, np.pi)]
            g_empty = scenes.build_grid((-0.75, -1.5, 0.0), (150, 150, 150), 0.02, 0.4, [])
            cfg_p = dataclasses.replace(cfg, name="pr2_empty_world", robot_text=pr2, grid=g_empty)
            sp = capi.Space.from_config(cfg_p, batch_states=256)
            Qp = scenes.benchmark_states(lim, n2, 12345)
            dq = torch.from_numpy(Qp).to(dev)
            dv = torch.zeros(n2, dtype=torch.uint8, device=dev)
            dl = torch.zeros(n2, dtype=torch.int32, device=dev)
            for _ in range(2):
                sp.state_valid_batch_device(dq.data_ptr(), n2, dv.data_ptr(), dl.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            e0.record(stream)
            for _ in range(reps):
                sp.state_valid_batch_device(dq.data_ptr(), n2, dv.data_ptr(), dl.data_ptr(), stream.cuda_stream)
            e1.record(stream)
            torch.cuda.synchronize()
            msp = e0.elapsed_time(e1) / reps
            kp = {"robot": "PR2 right arm from tests/golden (collision_model_pr2.yaml of the reference, URDF subset, right-arm rows of "
                           "the demo's allowed-collision matrix): 23 leaf spheres on 8 links, 5 checked link pairs",
                  "world": "empty 150^3 grid @ 0.02 m", "configs": n2, "kernel_ms": round(msp, 4),
                  "collision_checks_per_s": round(n2 / (msp * 1e-3), 1), "valid_fraction": round(int(dv.sum(dtype=torch.int64).item()) / n2, 4)}
            lkp = int(dl.sum(dtype=torch.int64).item())
            kp_bytes = 4.0 * lkp + 8.0 * N * n2
            kp_traffic = None
            try:
                with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                    kp_traffic = json.load(f)["k_state_valid_pr2"]["bytes_per_launch"] if n2 == (1 << 20) else None
            except (OSError, KeyError, ValueError):
                kp_traffic = None
            kp["lookups"] = lkp
            kp["roofline"] = {"bound": "hbm", "achieved": round(kp_bytes / (msp * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                              "frac": round(kp_bytes / (msp * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6), "traffic": kp_traffic}
            if Oracle is not None:
                o = Oracle(cfg_p)
                o.set_order(chain=True)
                ns = min(n2, 200000)
                ok_c, lk_c, sec = o.state_valid_batch_timed(Qp[:ns])
                kp["cpu_checks_per_s_one_core"] = round(ns / sec, 1)
                kp["parity"] = {"valid_bits_equal": bool(np.array_equal(ok_c, dv[:ns].cpu().numpy())),
                                "lookups_equal_on_valid": bool(np.array_equal(lk_c[ok_c == 1], dl[:ns].cpu().numpy()[ok_c == 1]))}
                del o
            out["k2_pr2"] = kp
            del dq, dv, dl, sp

    if single and Oracle is not None:
        o = Oracle(cfg)
        o.set_goal_joint(q_goal, cfg.goal_tol)
        r = o.eval_batch_timed(Q, args.cpu_seconds)
        cpu_rate = r["evals"] / r["seconds"]
        out["cpu_baseline"] = {"value": round(cpu_rate, 1), "unit": "successor evaluations/s", "cores": 1, "kind": "port",
                               "sample": f"{r['passes']} pass(es) over the same {B}-state frontier batch "
                                         f"({r['evals']} evaluations, {r['seconds']:.1f} s), oracle/ C++ -O2, 1 thread, "
                                         f"logging/visualisation off, 4-byte cells; host has {os.cpu_count()} cores"}
        # the same with the reference's own cell layout (48-byte array-of-structures cells over the padded grid,
        # distance_map.h:110-127; 824 MB at 256^3): SURVEY 8(d) asks for both
        o.use_aos_cells(True)
        r48 = o.eval_batch_timed(Q, min(args.cpu_seconds, 5.0))
        out["cpu_baseline"]["value_48_byte_cells"] = round(r48["evals"] / r48["seconds"], 1)
        out["cpu_baseline"]["sample_48_byte_cells"] = f"{r48['passes']} pass(es), {r48['evals']} evaluations, {r48['seconds']:.1f} s"
        del o

    if single and not args.no_planner:
        # ---- one query through the plugin API, run to its stated end: cfg 2's eps 5 ARA* until the goal is reached (first
        # solution; smpl_test/src/call_planner.cpp:1727-1729).  A lone query takes the host-driven loop (smplx_plan) ----
        nb = args.planner_expansions
        sp2 = new_space()
        sp2.set_goal_joint(q_goal, cfg.goal_tol)
        sp2.set_start(q_start)
        rg = sp2.plan(p.eps0, p.eps_final, p.eps_delta, False, True, nb, nb)
        out["planner"] = {
            "query": "cfg2 single query, ARA* eps 5 until the first solution (improve off), expansion bound %d, smplx_plan "
                     "(host-driven loop: sequential ARA* on the host, speculative frontier batches on the GPU)" % nb,
            "gpu_states_expanded_per_s": round(rg["expansions"] / rg["seconds"], 1),
            "succ_evals_per_s_committed": round(rg["committed_succ_evals"] / rg["seconds"], 1),
            "succ_evals_per_s_gpu_total": round(rg["gpu_succ_evals"] / rg["seconds"], 1),
            "solved": rg["solved"], "expansions": rg["expansions"], "path_cost": rg["cost"], "path_len": int(len(rg["path"])),
            "satisfied_eps": rg["satisfied_eps"], "states": sp2.num_states(),
            "gpu_seconds": round(rg["seconds"], 4),
            "gpu_succ_evals_total": rg["gpu_succ_evals"], "committed_succ_evals": rg["committed_succ_evals"],
            "gpu_batches": rg["gpu_batches"], "cache_hits": rg["cache_hits"], "cache_misses": rg["cache_misses"]}
        # a second timed frontier, deep in that search: the LAST B states it created (next to the obstacles the search has been
        # working around), so that joint limits, colliding edges and the sparse case of the compaction are inside a timed step
        nst = sp2.num_states()
        if nst > 2 * B:
            Qd = np.stack([sp2.get_state(i)[0] for i in range(nst - B, nst)])
            d_qd = torch.from_numpy(Qd).to(dev)
            cnt_d = torch.zeros_like(d_cnt)

            def step_deep():
                space.expand_batch_k5_device(d_qd.data_ptr(), B, d_flags.data_ptr(), d_coord.data_ptr(), d_sq.data_ptr(),
                                             d_h.data_ptr(), d_cost.data_ptr(), d_lk.data_ptr(), d_id.data_ptr(), d_reca.data_ptr(), cap_k5,
                                             d_recb.data_ptr(), cap_k5, d_btab.data_ptr(), d_tot.data_ptr(), d_work.data_ptr(),
                                             cnt_d.data_ptr(), stream.cuda_stream)
            for _ in range(5):
                step_deep()
            torch.cuda.synchronize()
            cnt_d.zero_()
            nd = max(args.steps, 200)
            td0 = time.perf_counter()
            for _ in range(nd):
                step_deep()
            torch.cuda.synchronize()
            td1 = time.perf_counter()
            ev_d, va_d, lkr_d, lkd_d, cf_d, sl_d = space.counters_read(cnt_d.data_ptr(), B)
            fl = d_flags.cpu().numpy()
            out["deep_frontier"] = {
                "states": f"the last {B} of the {nst} states the solved cfg-2 search created", "steps": nd,
                "ms_per_step": round(1e3 * (td1 - td0) / nd, 4), "successor_evaluations_per_s": round(ev_d / (td1 - td0), 1),
                "valid_fraction": round(va_d / max(ev_d, 1), 4),
                "edges_out_of_limits": int(((fl & 0x20) != 0).sum()), "edges_in_collision": int(((fl & 0x40) != 0).sum()),
                "edges_inactive": int(((fl & 0x10) != 0).sum()), "configs_per_launch": int(cf_d / nd)}
            del d_qd, cnt_d
        # the same query, 40 000 expansions, on the device-resident search (one workgroup: what one query of a shard gets)
        rd_log = {}
        os.environ["SMPLX_SEARCH"] = "device"
        try:
            sp3 = new_space()
            sp3.set_goal_joint(q_goal, cfg.goal_tol)
            sp3.set_start(q_start)
            rd = sp3.plan(p.eps0, p.eps_final, p.eps_delta, True, True, 40000, 40000)
            rd_log["expansion_log"] = rd["expansion_log"]
            sc3 = sp3.search_counters()
            tsum = sum(v for k_, v in sc3.items() if k_.startswith("t_")) or 1
            out["planner_device"] = {
                "query": "the cfg2 query, 40 000 expansions, device-resident ARA* (SMPLX_SEARCH=device): one persistent workgroup",
                "states_expanded_per_s": round(rd["expansions"] / rd["seconds"], 1), "expansions": rd["expansions"],
                "kernel_launches": rd["gpu_batches"], "seconds": round(rd["seconds"], 4),
                "workgroup_time_share": {k_[2:]: round(v / tsum, 3) for k_, v in sc3.items() if k_.startswith("t_") and v},
                "us_per_expansion": round(1e6 * rd["seconds"] / max(rd["expansions"], 1), 2)}
            del sp3
        except capi.SmplxError as e:
            out["planner_device"] = {"error": str(e)[:200]}
        finally:
            os.environ.pop("SMPLX_SEARCH", None)
        if Oracle is not None:
            o2 = Oracle(cfg)
            o2.set_goal_joint(q_goal, cfg.goal_tol)
            o2.set_start(q_start)
            o2.search_params(p.eps0, p.eps_final, p.eps_delta, False, True, nb, nb)
            ro = o2.plan()
            out["planner"].update({
                "cpu_states_expanded_per_s": round(ro["expansions"] / ro["seconds"], 1), "cpu_seconds": round(ro["seconds"], 4),
                "cpu_succ_evals_per_s": round(ro["succ_evals"] / ro["seconds"], 1),
                "parity": {"cost_equal": bool(ro["cost"] == rg["cost"]), "solved_equal": bool(ro["ok"] == rg["solved"]),
                           "expansions_equal": bool(ro["expansions"] == rg["expansions"]),
                           "expanded_ids_equal": bool(np.array_equal(ro["expansion_log"], rg["expansion_log"])),
                           "path_equal": bool(np.array_equal(ro["path"], rg["path"])),
                           "succ_evals_equal": bool(ro["succ_evals"] == rg["committed_succ_evals"])}})
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
                want, ec, sc_ = o2.post_process(P, True, True, True)
                tc = time.perf_counter() - t0
                out["planner"]["post_process"] = {
                    "points_in": int(len(P)), "points_out": int(len(got)), "gpu_ms": round(tg * 1e3, 3),
                    "cpu_ms": round(tc * 1e3, 3), "gpu_configs_checked": int(st["configs"]),
                    "gpu_batches": int(st["edge_batches"]), "cpu_edge_checks": int(ec), "cpu_state_checks": int(sc_),
                    "equal": bool(got.shape == want.shape and np.array_equal(got, want))}
            del o2
        del sp2

        # ---- the same query through an SBPL-shaped loop: plain GetSuccs / GetGoalHeuristic, no hints -----------------
        try:
            with tempfile.TemporaryDirectory() as td:
                from smpl_amd.plugin_tools import build_driver, write_query
                import copy
                exe = build_driver("sbpl_loop_driver", td)
                c2 = copy.copy(cfg)
                c2.start, c2.goal = list(q_start), list(q_goal)
                nbp = 40000
                write_query(c2, td, [p.eps0, p.eps_final, p.eps_delta, nbp, nbp])
                pr = subprocess.run([exe, td, "log"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
                lines = {l.split(" ", 1)[0]: l.split(" ", 1)[1] if " " in l else "" for l in pr.stdout.decode().splitlines()}
                if pr.returncode == 0 and "result" in lines:
                    solved, cost, nexp, plen, secs, eps = lines["result"].split()
                    st = dict(kv.split("=") for kv in lines.get("stats", "").split()) if "stats" in lines else {}
                    out["planner_plain"] = {
                        "caller": "SBPL-shaped ARA* (tests/cpp/sbpl_loop_driver.cpp) over include/smpl_amd/plugin.hpp; only "
                                  "GetSuccs / GetGoalHeuristic, no smplx_hint_frontier",
                        "states_expanded_per_s": round(int(nexp) / float(secs), 1), "expansions": int(nexp),
                        "path_cost": int(cost), "seconds": round(float(secs), 4), **{k: int(v) for k, v in st.items()},
                        "log_equals_device_search": bool("expansion_log" in rd_log and
                                                         np.array_equal(np.array(lines.get("log", "").split(), dtype=np.int64), rd_log["expansion_log"]))}
                else:
                    out["planner_plain"] = {"error": pr.stderr.decode()[-300:]}
        except Exception as e:   # the secondary leg must not take the bench line down
            out["planner_plain"] = {"error": repr(e)[:300]}

    if not args.no_shard and not args.no_planner:
        # ---- BASELINE config 4: this rank's 128 queries through smplx_plan_multi: the device-resident search, one persistent
        # workgroup per query, all in one launch ------------------------------------------------------------------------
        nb = args.shard_expansions

        def make_spaces():
            # goal (BFS_3D::run to completion) and start of every query; eight host threads, each space on its own stream, so
            # that the narrow passes of one BFS overlap with another's (a lone 256^3 BFS leaves most of the chip idle)
            from concurrent.futures import ThreadPoolExecutor
            t_ = time.perf_counter()

            def one(ab):
                sp = new_space(1024)
                sp.set_goal_joint(ab[1], cfg.goal_tol)
                sp.set_start(ab[0])
                return sp
            with ThreadPoolExecutor(max_workers=args.setup_threads, initializer=lambda: torch.cuda.set_device(dev_index)) as ex:
                sps = list(ex.map(one, zip(S_mine, G_mine)))
            torch.cuda.synchronize()
            return sps, time.perf_counter() - t_
        spaces, t_set = make_spaces()
        if dist is not None:
            dist.barrier()
        res, wall = ([], 0.0)
        if spaces:
            res, wall = capi.Space.plan_multi(spaces, p.eps0, p.eps_final, p.eps_delta, True, True, nb, nb,
                                              host_threads=args.host_threads)
        if dist is not None:
            dist.barrier()
        rec = shard.pack_records(first, res)
        rows = shard.gather_query_records(rec, args.queries_per_gpu, dist, world, coll_dev)    # the one collective of the path
        tot_exp = sum(r["expansions"] for r in res)
        tot_commit = sum(r["committed_succ_evals"] for r in res)
        tot_gpu = sum(r["gpu_succ_evals"] for r in res)
        sc2 = shard.gather_scalars([tot_exp, wall, tot_commit, tot_gpu, max([r["gpu_batches"] for r in res] or [0]), t_set], dist, world, coll_dev)
        tmax2 = float(sc2[:, 1].max())
        tset2 = float(sc2[:, 5].max())
        summ = shard.summarize(rows)
        on_device = bool(res) and all(r["cache_misses"] == 0 for r in res)
        out["shard"] = {
            "workload": f"cfg 4: queries [{first}, {last}) of the seeded list (seed 4) per rank, {args.queries_per_gpu} per GPU, "
                        f"ARA* eps 5->1, expansion bound {nb} per query, smplx_plan_multi",
            "search": "device-resident: one persistent workgroup per query, one launch for the shard, no host round trips"
                      if on_device else f"host-driven loop, {args.host_threads} worker threads + 1 GPU submitter",
            "queries": summ["queries"], "solved": summ["solved"], "wall_seconds_max": round(tmax2, 4),
            "states_expanded_per_s": round(float(sc2[:, 0].sum()) / tmax2, 1) if tmax2 > 0 else 0.0,
            "succ_evals_per_s_committed": round(float(sc2[:, 2].sum()) / tmax2, 1) if tmax2 > 0 else 0.0,
            "succ_evals_per_s_gpu_total": round(float(sc2[:, 3].sum()) / tmax2, 1) if tmax2 > 0 else 0.0,
            "expansions_total": summ["expansions_total"], "kernel_launches_or_batches": int(sc2[:, 4].max()),
            "setup_seconds_max": round(tset2, 3), "cost_checksum": summ["cost_checksum"],
            "setup_is": "goal + BFS_3D::run + start for every query of the rank (the GPU's BFS per goal), outside wall_seconds",
            "states_expanded_per_s_incl_setup": round(float(sc2[:, 0].sum()) / (tmax2 + tset2), 1) if tmax2 > 0 else 0.0}
        if single and spaces and on_device:
            sc0 = spaces[0].search_counters()
            tsum = sum(v for k_, v in sc0.items() if k_.startswith("t_")) or 1
            out["shard"]["workgroup_time_share_query0"] = {k_[2:]: round(v / tsum, 3) for k_, v in sc0.items() if k_.startswith("t_") and v}
            out["shard"]["us_per_expansion_query0"] = round(tsum * 0.01 / max(res[0]["expansions"], 1), 2)   # ticks of 10 ns
            # the same shard on the host-driven loop (what round 2 measured): secondary, A/B on the same box
            os.environ["SMPLX_SEARCH"] = "host"
            try:
                sp_h, _ = make_spaces()
                res_h, wall_h = capi.Space.plan_multi(sp_h, p.eps0, p.eps_final, p.eps_delta, True, True, nb, nb, host_threads=args.host_threads)
                out["shard_host_loop"] = {
                    "search": f"host-driven loop, {args.host_threads} worker threads + 1 GPU submitter (SMPLX_SEARCH=host)",
                    "states_expanded_per_s": round(sum(r["expansions"] for r in res_h) / wall_h, 1), "wall_seconds": round(wall_h, 4),
                    "gpu_batches": int(sum(r["gpu_batches"] for r in res_h)),
                    "identical_to_device_search": bool(all(a_["cost"] == b_["cost"] and a_["expansions"] == b_["expansions"] and
                                                           np.array_equal(a_["path"], b_["path"]) for a_, b_ in zip(res, res_h)))}
                del sp_h
            finally:
                os.environ.pop("SMPLX_SEARCH", None)
        if single and Oracle is not None and spaces:
            # the same queries on the oracle, one per host thread ("nproc independent queries in parallel,
            # one per core"); a bounded sample: every thread takes queries t, t+T, ... until the budget is spent.
            # Twice: with a GPU's share of the cores of an 8-GPU node (nproc // 8), and with all of them.
            def cpu_shard_run(T, budget):
                acc = [dict(exp=0, ev=0, plan_s=0.0, setup_s=0.0, n=0, same=True) for _ in range(T)]
                t_all = time.perf_counter()

                def worker(t):
                    k = t
                    while k < len(spaces) and time.perf_counter() - t_all < budget:
                        ts = time.perf_counter()
                        o = Oracle(cfg)               # a fresh context per query: ids restart at the goal (0) and the start (1)
                        o.set_goal_joint(G_mine[k], cfg.goal_tol)
                        o.set_start(S_mine[k])
                        acc[t]["setup_s"] += time.perf_counter() - ts
                        o.search_params(p.eps0, p.eps_final, p.eps_delta, True, True, nb, nb)
                        r = o.plan()
                        acc[t]["exp"] += r["expansions"]; acc[t]["ev"] += r["succ_evals"]; acc[t]["plan_s"] += r["seconds"]
                        acc[t]["n"] += 1
                        acc[t]["same"] &= bool(r["cost"] == res[k]["cost"] and r["expansions"] == res[k]["expansions"] and
                                               np.array_equal(r["expansion_log"], res[k]["expansion_log"]))
                        k += T
                th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
                for x in th:
                    x.start()
                for x in th:
                    x.join()
                busy = max(a_["plan_s"] for a_ in acc) or 1.0
                busy_all = max(a_["plan_s"] + a_["setup_s"] for a_ in acc) or 1.0
                nq_cpu = sum(a_["n"] for a_ in acc)
                return {
                    "value": round(sum(a_["exp"] for a_ in acc) / busy, 1), "unit": "states expanded/s", "cores": T, "kind": "port",
                    "succ_evals_per_s": round(sum(a_["ev"] for a_ in acc) / busy, 1),
                    "states_expanded_per_s_incl_setup": round(sum(a_["exp"] for a_ in acc) / busy_all, 1),
                    "bfs_and_start_seconds_per_query": round(sum(a_["setup_s"] for a_ in acc) / max(nq_cpu, 1), 3),
                    "sample": f"{nq_cpu} of the {len(spaces)} shard queries, one per host thread at a time ({T} threads, oracle/ "
                              f"C++ -O2, same expansion bound), rate = expansions of all threads / longest thread's time inside "
                              f"plan(); the _incl_setup figure adds BFS_3D::run + setStart per query on both sides",
                    "all_sampled_queries_identical_to_gpu": bool(all(a_["same"] for a_ in acc)), "host_cores": os.cpu_count()}
            ncpu = host_threads_available(1 << 20)
            if args.cpu_threads > 0:
                out["cpu_shard"] = cpu_shard_run(host_threads_available(args.cpu_threads), args.cpu_seconds)
            else:
                out["cpu_shard"] = cpu_shard_run(max(1, ncpu // 8), args.cpu_seconds)
                out["cpu_shard"]["cores_are"] = "nproc // 8: one GPU's share of the host cores of an 8-GPU node"
                if ncpu // 8 != ncpu:
                    out["cpu_shard_all_cores"] = cpu_shard_run(ncpu, args.cpu_seconds)
                    out["cpu_shard_all_cores"]["cores_are"] = "nproc: every host core of the box (SURVEY 8d)"
        del spaces

    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
