"""Per-kernel median durations of a rocprofv3 --kernel-trace output directory (developer tool):  python tools/kernel_medians.py <dir>"""
import csv, glob, sys
from collections import defaultdict
d = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[(r["Kernel_Name"][:100], r["Grid_Size_X"], r["Workgroup_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:20]:
    v.sort()
    print("%-102s grid %8s wg %4s n %3d median %8.1f us" % (k[0], k[1], k[2], len(v), v[len(v) // 2]))
