cd $GRAFT_REPO_ROOT
for cfg in "8 40" "8 32" "8 24" "8 16" "8 8" "6 32" "10 32"; do set -- $cfg
  export MI355RT_TRACE_REFILL=$1 MI355RT_TRACE_NODE_LANES=$2
  h=$(timeout -k 10 300 python tools/bench_hall.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['closest_grays_per_s_in_kernel'],3))")
  p=$(timeout -k 10 300 python tools/bench_ply_scene.py --no-oracle --spp 128 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['closest_grays_per_s_in_kernel'],3))")
  echo "refill $1 lanes $2: hall $h ply $p"
done
