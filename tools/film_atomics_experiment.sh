cd $GRAFT_REPO_ROOT
for v in "" filmstore filmnone; do
  if [ -n "$v" ]; then export MI355RT_LIB=$GRAFT_REPO_ROOT/ray_tracer_amd/libv/$v/libmi355rt.so; fi
  for alg in mis bdpt; do
    timeout -k 10 200 python bench.py --no-cpu --no-soup --no-soup10m --algorithm $alg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', '$alg', round(d['value'],1), {k: round(x/d['steps'],2) for k,x in d['kernels_ms'].items()})" || exit 1
  done
done
