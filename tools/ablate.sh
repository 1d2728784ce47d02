#!/bin/bash
# diagnostic: rebuild the library with one part of the collision kernel disabled and time the bench step
set -e
cd "$(dirname "$0")/.."
for V in ${ABL_LIST:-NONE ABL_NO_LOOKUP ABL_NO_SINCOS ABL_NO_PAIRS ABL_NO_TREES ABL_NO_FK}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -D$V smpl_amd/csrc/kernels.hip smpl_amd/csrc/engine.hip smpl_amd/csrc/model_compile.cpp -o smpl_amd/libsmpl_amd.so 2>/dev/null
  python bench.py --no-cpu --no-planner --overlap-streams 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['finish_kernel_ms'])"
done
