cd $GRAFT_REPO_ROOT
for cfg in "16 62" "16 56" "16 48" "12 62" "8 56" "8 48" "8 40" "8 32"; do set -- $cfg
  export MI355RT_TRACE_REFILL=$1 MI355RT_TRACE_NODE_LANES=$2
  timeout -k 10 300 python tools/bench_trace.py "refill $1 lanes $2" --soup10m 2>/dev/null | tail -1
done
