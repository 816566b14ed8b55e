#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + the two PMC passes of the default bench command.
#   usage: tools/profile_gpu.sh <tag>      -> gpurun_out/prof_<tag>/{trace,fetch,write}
set -e
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-events > $OUT/trace_bench.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-events > $OUT/fetch_bench.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-events > $OUT/write_bench.json 2> $OUT/write.err
echo "profiles in $OUT"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/mfma -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-events > $OUT/mfma_bench.json 2> $OUT/mfma.err || true
