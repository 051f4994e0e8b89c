#!/bin/bash
# builds a variant of the library into ray_tracer_amd/libv/<name>/ (kernel experiments; select it with MI355RT_LIB)
# usage: tools/build_variant.sh <name> "<extra compiler flags>"
set -e
cd "$(dirname "$0")/.."
OUT=$PWD/ray_tracer_amd/libv/$1
mkdir -p $OUT/obj
make -C ray_tracer_amd/csrc OUT=$OUT EXTRA="$2" $OUT/libmi355rt.so -j8 > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
echo built $OUT/libmi355rt.so
