#!/bin/bash
# Instruction-cache counters of one bench step per kernel (run on the GPU box): gpurun_out/prof_icache/
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_icache
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $OUT/ic -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-events > $OUT/ic_bench.json 2> $OUT/ic.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-events > $OUT/sq_bench.json 2> $OUT/sq.err
python3 - <<PY
import csv, glob, collections
for tag in ("ic", "sq"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
    print("==", tag)
    for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:14]:
        n = max(cnt[(k, c)] for c in d)
        print(f"{k:70s} launches {n:5d} " + " ".join(f"{c}={v/n:.0f}" for c, v in sorted(d.items())))
PY
