#!/bin/bash
# rocprofv3 evidence for ONE forward shape through the operator (developer tool, run from the repo root on the GPU box):
#   bash tools/prof_shape.sh <tag> B H N D f16|bf16 causal(0|1) [Nkv]
# kernel trace + separate --pmc passes (never combined with other trace domains); prints per-kernel medians and per-launch counters.
TAG=$1; B=$2; H=$3; N=$4; D=$5; DT=${6:-f16}; C=${7:-0}; NKV=${8:-$4}
OUT=gpurun_out/prof_${TAG}
export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
cat > /tmp/prof_shape.py <<P
import os, sys, torch
sys.path.insert(0, os.path.join("$PWD", "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import FlashAttentionFunction
dt = torch.float16 if "$DT" == "f16" else torch.bfloat16
q = torch.rand(($B, $H, $N, $D), device="cuda").to(dt)
k, v = (torch.rand(($B, $H, $NKV, $D), device="cuda").to(dt) for _ in range(2))
for _ in range(int(os.environ.get("CALLS", "60"))):
    FlashAttentionFunction.apply(q, k, v, None, bool($C))
torch.cuda.synchronize()
P
CALLS=200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python /tmp/prof_shape.py > $OUT/trace.log 2>&1
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d $OUT/p$i -o pmc -- python /tmp/prof_shape.py > $OUT/p$i.log 2>&1
done
OUT=$OUT SHAPE="B$B H$H N$N x $NKV D$D $DT causal=$C" python - <<'P'
import csv, glob, os
from collections import defaultdict
out = os.environ["OUT"]
print("== %s, FlashAttentionFunction.apply ==" % os.environ["SHAPE"])
dur = defaultdict(list)
for path in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "fa2::" in r["Kernel_Name"]:
            dur[(r["Kernel_Name"].split("(")[0].replace("void fa2::", ""), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, wgs), v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("%-50s workgroups %5d  launches %3d  median %8.1f us  min %8.1f  mean(last half) %8.1f" % (name[:50], wgs, len(v), v[len(v) // 2], v[0], sum(v[:len(v)//2]) / max(1, len(v)//2)))
acc = defaultdict(lambda: [0.0, 0])
for path in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        kn = r.get("Kernel_Name", "")
        if "fa2::" not in kn:
            continue
        key = (kn.split("(")[0].replace("void fa2::", ""), r["Counter_Name"])
        acc[key][0] += float(r["Counter_Value"] or 0); acc[key][1] += 1
res = defaultdict(dict)
for (kn, c), (s, n) in sorted(acc.items()):
    res[kn][c] = s / n
    print("%-50s %-28s %14.5g  (%d launches)" % (kn[:50], c, s / n, n))
for kn, r in res.items():
    if "GRBM_GUI_ACTIVE" in r and "SQ_VALU_MFMA_BUSY_CYCLES" in r:
        cyc = r["GRBM_GUI_ACTIVE"] / 8.0
        line = "%-50s cycles %9.0f  matrix pipe busy %.3f" % (kn[:50], cyc, r["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc))
        if "SQ_VALU_MFMA_COEXEC_CYCLES" in r: line += "  COEXEC/BUSY %.2f" % (r["SQ_VALU_MFMA_COEXEC_CYCLES"] / r["SQ_VALU_MFMA_BUSY_CYCLES"])
        if "SQ_INSTS_VALU" in r and "SQ_INSTS_MFMA" in r: line += "  VALU(non-MFMA)/MFMA %.2f" % ((r["SQ_INSTS_VALU"] - r["SQ_INSTS_MFMA"]) / r["SQ_INSTS_MFMA"])
        if "FETCH_SIZE" in r: line += "  HBM bytes (2*FETCH+WRITE)*1024 = %.4g" % ((2 * r["FETCH_SIZE"] + r.get("WRITE_SIZE", 0)) * 1024)
        print(line)
P
