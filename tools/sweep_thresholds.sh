# the traversal kernel's two launch-time thresholds (idle lanes that trigger a refill, descending lanes below which the node phase ends)
# for an integrator's rays in larger scenes: hall and PLY scene, closest-hit rate inside the kernel
cd $GRAFT_REPO_ROOT
for cfg in "8 40" "8 48" "8 32" "4 40" "12 40" "16 40" "8 36" "8 44"; do set -- $cfg
  export MI355RT_TRACE_REFILL=$1 MI355RT_TRACE_NODE_LANES=$2
  h=$(timeout -k 10 300 python tools/bench_hall.py --spp 64 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['spp_per_s'],1), round(d['closest_grays_per_s_in_kernel'],3))") || exit 1
  p=$(timeout -k 10 300 python tools/bench_ply_scene.py --no-oracle 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['spp_per_s'],1), round(d['closest_grays_per_s_in_kernel'],3))") || exit 1
  echo "refill $1 lanes $2: hall $h ply $p"
done
