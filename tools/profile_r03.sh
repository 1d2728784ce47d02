#!/bin/bash
# Round-3 profiles on the GPU box, csv under gpurun_out/r03_prof/ (tools/summarize_r03.py condenses them for profiles/):
#   round/   the bench step: kernel trace + FETCH_SIZE / WRITE_SIZE / SQ counters in their own passes (tools/profile_round.sh)
#   k2*      the collision micro-benchmarks (bench.py k2, k2_pr2, random_frontier legs): kernel trace, FETCH_SIZE
#   bfs*     BFS per goal at 256^3 and 512^3: kernel trace, FETCH_SIZE, WRITE_SIZE (tools/profile_bfs.sh)
#   search*  the device-resident ARA* on the cfg-4 shard (tools/search_probe.py): kernel trace
# rocprofv3 gets python3 directly after "--".
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_prof
mkdir -p "$OUT"
tools/profile_round.sh r03_prof/round > "$OUT/round.log" 2>&1
K2="bench.py --no-cpu --no-planner --overlap-streams 1 --scaling-batches= --steps 10"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o k2 -- python3 $K2 > "$OUT/k2.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT" -o k2fetch -- python3 $K2 > "$OUT/k2fetch.log" 2>&1
tools/profile_bfs.sh r03_prof/bfs256 256 > "$OUT/bfs256.log" 2>&1
tools/profile_bfs.sh r03_prof/bfs512 512 > "$OUT/bfs512.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o search -- python3 tools/search_probe.py 20000 128 > "$OUT/search.log" 2>&1
find "$OUT" -name "*.csv" | wc -l
