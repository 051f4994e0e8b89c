cd $GRAFT_REPO_ROOT
for x in 0 1; do
  export MI355RT_TRACE_TWO=$x
  b=$(timeout -k 10 200 python bench.py --no-cpu --no-soup 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['kernels_ms']['closest'],1), d['rays_per_sample'])")
  h=$(timeout -k 10 300 python tools/bench_hall.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['spp_per_s'],1), round(d['closest_grays_per_s_in_kernel'],3))")
  p=$(timeout -k 10 300 python tools/bench_ply_scene.py --no-oracle --spp 128 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['spp_per_s'],1), round(d['closest_grays_per_s_in_kernel'],3))")
  echo "one launch for connection rays and next path rays: $x: bench $b | hall $h | ply $p"
done
