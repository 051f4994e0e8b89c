#!/bin/bash
# builds the library with the diagnostic counters into a variant directory and prints loop statistics and wave end times for the Cornell
# render rays, the hall's render rays and the 1M soup
cd $GRAFT_REPO_ROOT
bash tools/build_variant.sh stats "-DTRACE_STATS $1" || exit 1
export MI355RT_LIB=$PWD/ray_tracer_amd/libv/stats/libmi355rt.so
timeout -k 10 120 python3 tools/dev_stats.py cornell || exit 1
timeout -k 10 300 python3 tools/dev_stats.py hall || exit 1
timeout -k 10 120 python3 tools/dev_stats.py soup 1000000 0.004 4194304 || exit 1
