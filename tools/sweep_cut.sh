# the scenes with the cutting of long triangles on (default) and off: hall, hall with long polygons, PLY scene, soups
cd $GRAFT_REPO_ROOT
for p in 0 0.25; do
  export MI355RT_PRESPLIT=$p
  for f in "" "--long-polygons"; do
    timeout -k 10 300 python tools/bench_hall.py --spp 64 $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('budget $p', d['scene'], d['cut_triangles'], d['extra_references'], round(d['spp_per_s'],1), round(d['closest_grays_per_s_in_kernel'],3), 'upload', round(d['obj_load_s'],2))" || exit 1
  done
  timeout -k 10 300 python tools/bench_ply_scene.py --no-oracle --spp 128 --pass-spp 128 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('budget $p ply', round(d['spp_per_s'],1), round(d['closest_grays_per_s_in_kernel'],3))" || exit 1
  timeout -k 10 300 python tools/bench_trace.py cut$p --soup10m 2>/dev/null | tail -1 || exit 1
done
