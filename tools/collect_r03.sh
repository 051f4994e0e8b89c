# everything the round's record needs, on the final library: bench, integrators, PLY scenes, hall, 2-rank rehearsal, then the
# rocprofv3 passes (kernel trace + PMC) over the bench, the hall and the two soups
cd $GRAFT_REPO_ROOT
bash tools/collect_round_numbers.sh > gpurun_out/collect_r03.log 2>&1
bash tools/profile.sh r3_bench bench.py --no-cpu --no-soup > gpurun_out/r3_bench_profile.log 2>&1
bash tools/profile.sh r3_hall tools/bench_hall.py --spp 16 > gpurun_out/r3_hall_profile.log 2>&1
bash tools/profile.sh r3_soup1m tools/soup_trace.py 1000000 0.004 > gpurun_out/r3_soup1m_profile.log 2>&1
bash tools/profile.sh r3_soup10m tools/soup_trace.py 10000000 0.002 > gpurun_out/r3_soup10m_profile.log 2>&1
cat gpurun_out/r3_*/status.txt | sort | uniq -c
