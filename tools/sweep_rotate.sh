# device-built trees with and without rotations in the bottom-up pass (MI355RT_ROTATE): hall, hall with long polygons, PLY scene, soups
cd $GRAFT_REPO_ROOT
for r in 0 1 0 1; do
  export MI355RT_ROTATE=$r
  for f in "" "--long-polygons"; do
    timeout -k 10 300 python tools/bench_hall.py --spp 64 $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rotate $r', d['scene'], round(d['spp_per_s'],1), round(d['closest_grays_per_s_in_kernel'],3), 'build', round(d['bvh_build_s']*1e3,1), 'ms depth', d['bvh_depth'])" || exit 1
  done
  timeout -k 10 300 python tools/bench_ply_scene.py --no-oracle --spp 128 --pass-spp 128 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rotate $r ply', round(d['spp_per_s'],1), round(d['closest_grays_per_s_in_kernel'],3))" || exit 1
  timeout -k 10 300 python tools/bench_trace.py rot$r --soup10m 2>/dev/null | tail -1 || exit 1
done
