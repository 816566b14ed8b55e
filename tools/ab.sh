#!/bin/bash
# same-box A/B of two builds: tools/ab.sh <other.so> [bench args]   (the in-tree library is "new")
other=$1; shift
for i in 1 2; do
  for lib in new $other; do
    if [ $lib = new ]; then unset MTTS_HIP_LIB; else export MTTS_HIP_LIB=$PWD/$lib; fi
    python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$lib', 'ms/step', d['ms_per_step'], 'gemm ms', r['gemm_ms_per_step'], 'attn', r['attention']['ms_per_step'], 'elem', r['elementwise_ms_per_step'])"
  done
done
