#!/bin/bash
# A/B on one box: generic kernels vs the per-robot hiprtc build (bench step, B=4096)
B="python bench.py --no-cpu --no-planner --multi-queries 0 --steps 50"
pick='import json,sys;d=json.loads(sys.stdin.readline());print(sys.argv[1],d["kernels"][:30],d["ms_per_step"],d["roofline"]["kernel_ms"],d["roofline"]["finish_kernel_ms"])'
$B --generic-kernels | python -c "$pick" generic
$B | python -c "$pick" hiprtc
