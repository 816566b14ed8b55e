#!/bin/bash
# per-batch-size bench lines (non-default workloads, labelled as such in the JSON)
for b in "$@"; do
  python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c '
import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
print("B", d["config"]["per_gpu_batch"], "fr/s", d["value"], "ms/step", d["ms_per_step"], "gemm TF", r["achieved"], "gemm ms", r["gemm_ms_per_step"], "attn ms", r["attention"]["ms_per_step"], "elem ms", r["elementwise_ms_per_step"], "ms with events", r["ms_per_step_with_events"])'
done
