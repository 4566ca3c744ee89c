#!/bin/bash
# per-kernel time of the backward launches (developer tool)
export TMPDIR=/tmp
mkdir -p gpurun_out/bwdprof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bwdprof -o t -- python tools/gpu_check_bwd.py > gpurun_out/bwdprof/log.txt 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/bwdprof/t_kernel_stats.csv")))
for r in rows:
    if "fa2::" in r["Name"]:
        print("%-90s calls %4s avg %10.1f us  min %10.1f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
