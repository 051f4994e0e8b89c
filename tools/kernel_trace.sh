#!/bin/bash
# one rocprofv3 --kernel-trace --stats pass over a python command; table under gpurun_out/<tag>/summary.txt
# usage: tools/kernel_trace.sh <tag> <script.py and args...>     (run from the repository root on the GPU box)
set -u
TAG=$1; shift
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD=("$@")
for i in "${!CMD[@]}"; do case "${CMD[$i]}" in *.py) CMD[$i]="$REPO/${CMD[$i]}";; esac; done
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "${CMD[@]}" > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
cd $REPO
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
head -${LINES_SHOWN:-24} $OUT/summary.txt
