#!/bin/bash
# builds the library with the diagnostic counters and prints loop statistics for the hall scene's render rays
cd $GRAFT_REPO_ROOT
touch ray_tracer_amd/csrc/trace.hip
make -C ray_tracer_amd/csrc EXTRA="-DTRACE_STATS $1" > /dev/null 2>&1 || { echo BUILD FAILED; exit 1; }
timeout -k 10 300 python3 tools/dev_stats.py hall || exit 1
