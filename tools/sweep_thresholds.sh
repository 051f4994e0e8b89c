cd $GRAFT_REPO_ROOT
for cfg in "8 56" "8 48" "8 40" "4 52" "12 52" "8 52" "6 56"; do set -- $cfg
  export MI355RT_TRACE_REFILL=$1 MI355RT_TRACE_NODE_LANES=$2
  h=$(timeout -k 10 300 python tools/bench_hall.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['closest_grays_per_s_in_kernel'],3))")
  p=$(timeout -k 10 300 python tools/bench_ply_scene.py --no-oracle --spp 128 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['closest_grays_per_s_in_kernel'],3))")
  echo "refill $1 lanes $2: hall $h ply $p"
done
