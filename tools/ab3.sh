#!/bin/bash
# same-box comparison of several builds: tools/ab3.sh "<lib1> <lib2> ..." [bench args]; "new" = the in-tree library
libs=$1; shift
for i in 1 2; do
  for lib in $libs; do
    if [ $lib = new ]; then unset MTTS_HIP_LIB; else export MTTS_HIP_LIB=$PWD/$lib; fi
    python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$lib', 'ms/step', d['ms_per_step'], 'gemm ms', r['gemm_ms_per_step'], 'attn', r['attention']['ms_per_step'], 'elem', r['elementwise_ms_per_step'])"
  done
done
