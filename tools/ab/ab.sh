# A/B of library builds on one box: bash tools/ab/ab.sh name1 name2 ...  (tools/ab/<name>.so)
set -e
for round in 1 2; do
for v in "$@"; do
  MTTS_HIP_LIB=$PWD/tools/ab/$v.so timeout -k 10 200 python bench.py --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$v', d['ms_per_step'], 'gemm', r['gemm_ms_per_step'], 'attn', r['attention']['ms_per_step'], 'elem', r['elementwise_ms_per_step'])"
done
done
