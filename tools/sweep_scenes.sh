# usage: tools/sweep_scenes.sh <variant> ...   ("base" = the library in lib/): hall, PLY scene and the 10M soup per variant
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = "base" ]; then unset MI355RT_LIB; else export MI355RT_LIB=$PWD/ray_tracer_amd/libv/$v/libmi355rt.so; fi
  h=$(timeout -k 10 300 python tools/bench_hall.py --spp 64 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['spp_per_s'],1), round(d['closest_grays_per_s_in_kernel'],3))") || exit 1
  p=$(timeout -k 10 300 python tools/bench_ply_scene.py --no-oracle --spp 128 --pass-spp 128 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['spp_per_s'],1), round(d['closest_grays_per_s_in_kernel'],3))") || exit 1
  echo "variant $v: hall $h | ply $p"
  timeout -k 10 300 python tools/bench_trace.py $v --soup10m 2>/dev/null | tail -1 || exit 1
done
