#!/bin/bash
# GPU box: phase times of one brick visit of the BFS (k_bfs_brick_wave), from a diagnostic build of the library
# (-DSMPLX_BFS_TRACE: block 0 leaves the wall clock at the phase boundaries of its first brick; one line per pass).
# Usage: tools/bfs_trace.sh [256|512]   -> gpurun_out/bfs_trace_<n>.log
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
N=${1:-256}
mkdir -p gpurun_out /tmp/smplx_trace
(cd smpl_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -DSMPLX_BFS_TRACE -fgpu-rdc \
    kernels.hip engine.hip field.hip model_compile.cpp specialize.cpp -o /tmp/smplx_trace/libsmpl_amd_trace.so -L/opt/rocm/lib -lhiprtc -Wl,-rpath,/opt/rocm/lib)
SMPL_AMD_LIB=/tmp/smplx_trace/libsmpl_amd_trace.so SMPLX_DEBUG_TIMING=1 timeout -k 10 300 python3 tools/bfs_time.py "$N" > "gpurun_out/bfs_trace_$N.log" 2>&1
grep -c "pass" "gpurun_out/bfs_trace_$N.log"
