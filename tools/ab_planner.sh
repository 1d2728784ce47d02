#!/bin/bash
# planner leg A/B: generic kernels vs per-robot build (single query, cfg2, 40000 expansions)
B="python bench.py --steps 5 --cpu-seconds 1 --multi-queries 0 --overlap-streams 1"
pick='import json,sys;d=json.loads(sys.stdin.readline());p=d["planner"];print(sys.argv[1],d["kernels"][:40],p["gpu_states_expanded_per_s"],p["gpu_seconds"],p["gpu_batches"])'
$B --generic-kernels 2>/dev/null | python -c "$pick" generic
$B 2>/dev/null | python -c "$pick" per-robot
