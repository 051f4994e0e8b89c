set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err
for a in naive nee mis bdpt; do timeout -k 10 200 python bench.py --no-cpu --no-soup --algorithm $a --steps 6 --warmup 1 > $O/alg_$a.json 2>> $O/alg.err; done
timeout -k 10 300 python tools/bench_ply_scene.py --material glass > $O/ply_glass.json 2> $O/ply.err
timeout -k 10 300 python tools/bench_ply_scene.py --material gold > $O/ply_gold.json 2>> $O/ply.err
BENCH_REHEARSAL=1 timeout -k 10 400 python bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu > $O/rehearsal.out 2> $O/rehearsal.err

timeout -k 10 300 python tools/bench_hall.py --spp 64 > gpurun_out/final/hall.json 2> gpurun_out/final/hall.err
timeout -k 10 300 python tools/bench_hall.py --spp 64 --long-polygons > gpurun_out/final/hall_long.json 2>> gpurun_out/final/hall.err
MI355RT_PRESPLIT=0 timeout -k 10 300 python tools/bench_hall.py --spp 64 --long-polygons > gpurun_out/final/hall_long_whole.json 2>> gpurun_out/final/hall.err
echo collected
