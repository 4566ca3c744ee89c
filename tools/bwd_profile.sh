#!/bin/bash
# per-kernel time of the backward launches (developer tool): rocprofv3 kernel trace of tools/gpu_check_bwd.py, per (kernel, grid size) —
# an average over all launches of a kernel would mix the config-2 launches with the tiny shapes of the same instantiation
export TMPDIR=/tmp
mkdir -p gpurun_out/bwdprof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bwdprof -o t -- python tools/gpu_check_bwd.py > gpurun_out/bwdprof/log.txt 2>&1
python - <<'PY'
import csv
from collections import defaultdict
acc = defaultdict(list)
for r in csv.DictReader(open("gpurun_out/bwdprof/t_kernel_trace.csv")):
    if "fa2::" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, wgs), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("%-64s workgroups %6d  launches %3d  median %9.1f us  min %9.1f" % (name[:64], wgs, len(v), v[len(v) // 2], v[0]))
PY
