#!/bin/bash
# diagnostic: per-robot (hiprtc) build with one part of the collision kernel disabled; times the bench step.
# Results of these builds are wrong on purpose (parts are skipped); they only price the parts.
cd "$(dirname "$0")/.."
export SMPLX_CACHE_DIR=${SMPLX_CACHE_DIR:-/tmp/smplx_ablate_cache}
for V in ${ABL_LIST:-NONE ABL_NO_LOOKUP ABL_NO_SINCOS ABL_NO_PAIRS ABL_NO_TREES ABL_NO_FK}; do
  SMPLX_RTC_DEFINES="-D$V" python3 bench.py --no-cpu --no-planner --no-shard --no-k2 --overlap-streams 1 --scaling-batches= --steps 50 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['finish_kernel_ms'], d['kernels'][:12])"
done
