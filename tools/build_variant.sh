#!/bin/bash
# Build a variant of the library for same-box A/B runs: tools/build_variant.sh <name> "<extra hipcc flags>" <file.hip> [...]
# The named translation units are recompiled with the extra flags, everything else comes from matcha-tts-24k_amd/build/*.o;
# result: tools/ab/<name>.so (use with MTTS_HIP_LIB=$PWD/tools/ab/<name>.so).
set -e
name=$1; extra=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/matcha-tts-24k_amd
mkdir -p $R/tools/ab/obj_$name
objs=""
for o in $P/build/*.o; do
  b=$(basename $o .o); skip=0
  for f in "$@"; do [ "$(basename $f .hip)" = "$b" ] && skip=1; done
  [ $skip = 0 ] && objs="$objs $o"
done
for f in "$@"; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC $extra -c $P/csrc/$b.hip -o $R/tools/ab/obj_$name/$b.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $R/tools/ab/obj_$name/*.o -o $R/tools/ab/$name.so
echo built tools/ab/$name.so
