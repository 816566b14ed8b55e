#!/bin/bash
# same-box comparison of env settings: tools/env_ab2.sh "A=1,B=2 A=3 ..." [bench args]; a setting is a comma-separated list of VAR=value
sets=$1; shift
for i in 1 2; do
  for kv in $sets; do
    env $(echo $kv | tr ',' ' ') python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$kv', 'ms/step', d['ms_per_step'], 'gemm ms', r['gemm_ms_per_step'], 'attn', r['attention']['ms_per_step'], 'elem', r['elementwise_ms_per_step'], 'launches', r['launches_per_step'])"
  done
done
