#!/bin/bash
# builds the library with the diagnostic counters and prints loop statistics for the Cornell render rays and the 1M soup
cd $GRAFT_REPO_ROOT
touch ray_tracer_amd/csrc/trace.hip
make -C ray_tracer_amd/csrc EXTRA="-DTRACE_STATS $1" > /dev/null 2>&1 || { echo BUILD FAILED; exit 1; }
timeout -k 10 120 python3 tools/dev_stats.py cornell || exit 1
timeout -k 10 120 python3 tools/dev_stats.py soup 1000000 0.004 4194304 || exit 1
