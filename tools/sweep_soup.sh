#!/bin/bash
# builds the library with each EXTRA flag set given and times the closest-hit kernel on the 1M and 10M soups
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  touch ray_tracer_amd/csrc/trace.hip
  make -C ray_tracer_amd/csrc EXTRA="$cfg" > /dev/null 2>&1 || { echo "BUILD FAILED: $cfg"; continue; }
  echo "== [$cfg]"
  timeout -k 10 200 python3 tools/dev_soup.py 1000000 0.004 16777216 2>/dev/null | grep closest | tail -1
  MI355RT_BUILDER=device timeout -k 10 300 python3 tools/dev_soup.py 10000000 0.002 16777216 2>/dev/null | grep closest | tail -1
done
touch ray_tracer_amd/csrc/trace.hip
make -C ray_tracer_amd/csrc > /dev/null 2>&1
