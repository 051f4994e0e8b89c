# ordering of a large unordered batch: by the cell of the origin (the library), and by cell and direction octant (-DRAY_BIN_OCTANTS)
cd $GRAFT_REPO_ROOT
for v in "" octants; do  # the variant: a library built with 512 bins (cell * 8 + octant); not kept in the tree
  if [ -n "$v" ]; then export MI355RT_LIB=$GRAFT_REPO_ROOT/ray_tracer_amd/libv/$v/libmi355rt.so; fi
  timeout -k 10 300 python tools/bench_trace.py "bins:$v" --soup10m 2>/dev/null | tail -1 || exit 1
done
