#!/bin/bash
# builds the library with each EXTRA flag set given (one per argument) on the GPU box and times the soup trace
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  touch ray_tracer_amd/csrc/trace.hip
  make -C ray_tracer_amd/csrc EXTRA="$cfg" > /dev/null 2>&1 || { echo "BUILD FAILED: $cfg"; continue; }
  echo "== $cfg"
  timeout -k 10 200 python -u tools/dev_soup.py ${SOUP_N:-1000000} ${SOUP_SIGMA:-0.004} ${SOUP_RAYS:-16777216} 2>&1 | grep -E "closest|hit rate" | tail -2
done
touch ray_tracer_amd/csrc/trace.hip
make -C ray_tracer_amd/csrc > /dev/null 2>&1
