#!/bin/bash
# Profiles the default bench step on the GPU box: one kernel-trace pass and separate PMC passes (FETCH_SIZE,
# WRITE_SIZE, SQ counters), csv output under gpurun_out/$1, then tools/summarize_profile.py writes the summary.
# rocprofv3 gets python3 directly after "--" (no wrapper processes).
set -e
OUT=gpurun_out/${1:-prof_round}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="bench.py --no-cpu --no-planner --no-shard --no-k2 --overlap-streams 1 --scaling-batches= --steps 50"
python3 $ARGS > "$OUT/bench_line.json" 2> "$OUT/bench.err"      # also warms the on-disk kernel cache
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- python3 $ARGS > "$OUT/trace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT" -o fetch -- python3 $ARGS > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT" -o write -- python3 $ARGS > "$OUT/write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT" -o sq1 -- python3 $ARGS > "$OUT/sq1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT" -o sq2 -- python3 $ARGS > "$OUT/sq2.log" 2>&1
python3 tools/summarize_profile.py "$OUT" "$OUT/summary.json" "$OUT/kernel_stats.csv"
