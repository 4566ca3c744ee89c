"""Median duration per (kernel, grid) of a rocprofv3 --kernel-trace directory, fa2 kernels only (developer tool):  python tools/kernel_seq_medians.py <dir>"""
import csv, glob, sys
from collections import defaultdict
agg = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fa2::" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0].replace("void fa2::", "")[:60], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items()):
    v.sort()
    print("%-62s workgroups %6d  n %3d  median %8.1f us" % (k[0], k[1], len(v), v[len(v) // 2]))
