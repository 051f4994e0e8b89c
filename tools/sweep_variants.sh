# usage: tools/sweep_variants.sh <variant> ...   ("" = the library in lib/): bench, hall and PLY scene per variant
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = "base" ]; then unset MI355RT_LIB; else export MI355RT_LIB=$PWD/ray_tracer_amd/libv/$v/libmi355rt.so; fi
  b=$(timeout -k 10 200 python bench.py --no-cpu --no-soup 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['kernels_ms']['closest'],1))")
  h=$(timeout -k 10 300 python tools/bench_hall.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['spp_per_s'],1), round(d['closest_grays_per_s_in_kernel'],3))")
  p=$(timeout -k 10 300 python tools/bench_ply_scene.py --no-oracle --spp 128 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['spp_per_s'],1), round(d['closest_grays_per_s_in_kernel'],3))")
  echo "variant $v: bench $b | hall $h | ply $p"
  timeout -k 10 300 python tools/bench_trace.py $v --soup10m 2>/dev/null | tail -1
done
