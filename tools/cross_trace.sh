#!/bin/bash
# Kernel durations of the SDXL cross-attention forward (B2 H10 N4096 x Nkv 77, D64 fp16), this operator and torch SDPA, under rocprofv3
# --kernel-trace (developer tool, run from the repo root on the GPU box):  bash tools/cross_trace.sh [out-dir]
OUT=${1:-gpurun_out/cross_trace}
export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
cat > /tmp/cross_trace.py <<P
import os, sys, torch
sys.path.insert(0, os.path.join("$PWD", "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import FlashAttentionFunction
q = torch.rand((2, 10, 4096, 64), device="cuda").half()
k, v = (torch.rand((2, 10, 77, 64), device="cuda").half() for _ in range(2))
for _ in range(300):
    FlashAttentionFunction.apply(q, k, v, None, False)
torch.cuda.synchronize()
for _ in range(300):
    torch.nn.functional.scaled_dot_product_attention(q, k, v)
torch.cuda.synchronize()
P
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python /tmp/cross_trace.py > $OUT/trace.log 2>&1
OUT=$OUT python - <<'P'
import csv, glob, os
from collections import defaultdict
dur, gaps, last = defaultdict(list), defaultdict(list), {}
rows = []
for path in glob.glob(os.environ["OUT"] + "/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev_end, prev_name = None, None
for r in rows:
    name = r["Kernel_Name"].split("(")[0][:60]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[name].append((e - s) / 1e3)
    if prev_name == name:
        gaps[name].append((s - prev_end) / 1e3)
    prev_end, prev_name = e, name
for name, v in sorted(dur.items(), key=lambda kv: -len(kv[1]))[:4]:
    v.sort(); g = sorted(gaps[name]) or [0]
    print("%-62s launches %4d  kernel median %6.2f us  min %6.2f   gap to the next launch median %6.2f us" % (name, len(v), v[len(v) // 2], v[0], g[len(g) // 2]))
P
