#!/bin/bash
# wall time of the C++ host on the procedural hall: parse, flatten, upload, tree, one tiny render (run on the GPU box from the repository root)
set -e
cd $GRAFT_REPO_ROOT
python3 - <<'P'
import sys
sys.path.insert(0, '.')
from ray_tracer_amd import synth
print(synth.atrium('/tmp/hall_cli'))
P
for i in 1 2; do
  t0=$(date +%s%N)
  MI355RT_DEBUG=1 timeout -k 10 120 ray_tracer_amd/lib/rt_cli --obj /tmp/hall_cli/model.obj --materials /tmp/hall_cli/materials.json \
    --objmatmap /tmp/hall_cli/objmatmap.json --width 64 --height 64 --sample 1 --film /tmp/hall_cli/film.f64 2>&1 | grep -v amdgpu | tail -8
  echo "rt_cli hall 64x64x1: $(( ($(date +%s%N) - t0) / 1000000 )) ms wall"
done
