#!/bin/bash
# builds the library with each EXTRA flag set given (render.hip only) and runs the render benchmark (no CPU leg, no soup; BENCH_ARGS adds arguments, e.g. --algorithm bdpt)
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  touch ray_tracer_amd/csrc/render.hip
  make -C ray_tracer_amd/csrc EXTRA="$cfg" > /dev/null 2>&1 || { echo "BUILD FAILED: $cfg"; continue; }
  timeout -k 10 300 python -u bench.py --steps ${STEPS:-16} --warmup 2 --no-soup --no-cpu ${BENCH_ARGS:-} 2>/dev/null | python3 tools/show_bench.py "[$cfg]"
done
touch ray_tracer_amd/csrc/render.hip
make -C ray_tracer_amd/csrc > /dev/null 2>&1
