#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/p16_ablate
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R && rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 tools/p16_ablate.py > $OUT/cases.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000 for r in rows if "gemm_p16_kernel" in r["Kernel_Name"]]
cases = [l.strip()[5:] for l in open("$OUT/cases.txt") if l.startswith("CASE")]
for i, c in enumerate(cases):
    print(f"{c:40s} {min(d[3 * i:3 * i + 3]):8.1f} us")
PY
