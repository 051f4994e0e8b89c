# the hall with long polygons under other constants of the cutting rule (presplit.cpp): worth of a piece, misses per hit
cd $GRAFT_REPO_ROOT
for w in 1 4 16 64; do for r in 0.5 2 8; do
  MI355RT_CUT_WORTH=$w MI355RT_CUT_RATIO=$r timeout -k 10 300 python tools/bench_hall.py --spp 64 --long-polygons 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('worth $w ratio $r:', d['cut_triangles'], d['extra_references'], round(d['spp_per_s'],1), round(d['closest_grays_per_s_in_kernel'],3))" || exit 1
done; done
