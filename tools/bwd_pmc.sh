#!/bin/bash
# Backward kernels under rocprofv3 at one BASELINE shape (developer tool, run from the repo root on the GPU box):
#   bash tools/bwd_pmc.sh [c2|c3|c4|sdxl|d64] [out-dir]
# One kernel-trace pass (per (kernel, grid) median / min of the launch durations: an average over all launches of an instantiation
# would mix shapes) and separate --pmc passes (never combined with other trace domains), summarised on stdout.
CFG=${1:-c2}
OUT=${2:-gpurun_out/bwdpmc_$CFG}
export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
cat > /tmp/bwd_prof.py <<P
import os, sys, torch
sys.path.insert(0, os.path.join("$PWD", "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import FlashAttentionFunction
B, H, N, D, dt, causal = {"c2": (2, 16, 4096, 128, torch.float16, False), "c3": (2, 16, 4096, 128, torch.bfloat16, True),
                          "c4": (1, 32, 8192, 128, torch.float16, True), "sdxl": (2, 10, 4096, 64, torch.float16, False),
                          "d64": (2, 16, 4096, 64, torch.float16, False)}["$CFG"]
g = torch.Generator(device="cuda").manual_seed(1)
q, k, v = (torch.rand((B, H, N, D), generator=g, device="cuda").to(dt).requires_grad_(True) for _ in range(3))
o = FlashAttentionFunction.apply(q, k, v, None, causal)
go = torch.rand(o.shape, generator=g, device="cuda").to(dt)
for _ in range(int(os.environ.get("BWD_CALLS", "12"))):
    o.backward(go, retain_graph=True)
torch.cuda.synchronize()
P
BWD_CALLS=40 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python /tmp/bwd_prof.py > $OUT/trace.log 2>&1
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES" \
           "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d $OUT/p$i -o pmc -- python /tmp/bwd_prof.py > $OUT/p$i.log 2>&1
done
OUT=$OUT python - <<'P'
import csv, glob, os
from collections import defaultdict
out = os.environ["OUT"]
dur = defaultdict(list)
for path in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "fa2::bwd" in r["Kernel_Name"]:
            key = (r["Kernel_Name"].split("(")[0].replace("void fa2::", ""), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))
            dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("== kernel trace: per (kernel, workgroups) ==")
for (name, wgs), v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("%-48s workgroups %6d  launches %3d  median %8.1f us  min %8.1f  mean %8.1f" % (name[:48], wgs, len(v), v[len(v) // 2], v[0], sum(v) / len(v)))
acc = defaultdict(lambda: [0.0, 0])
for path in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        kn = r.get("Kernel_Name", "")
        if "fa2::bwd" not in kn:
            continue
        key = (kn.split("(")[0].replace("void fa2::", ""), r["Counter_Name"])
        acc[key][0] += float(r["Counter_Value"] or 0); acc[key][1] += 1
print("== PMC per launch ==")
res = defaultdict(dict)
for (kn, c), (s, n) in sorted(acc.items()):
    res[kn][c] = s / n
    print("%-48s %-28s %14.5g  (%d launches)" % (kn[:48], c, s / n, n))
print("== derived ==")
for kn, r in res.items():
    if "GRBM_GUI_ACTIVE" in r and "SQ_VALU_MFMA_BUSY_CYCLES" in r:
        cyc = r["GRBM_GUI_ACTIVE"] / 8.0
        line = "%-48s cycles %9.0f  matrix pipe busy %.3f" % (kn[:48], cyc, r["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc))
        if "SQ_VALU_MFMA_COEXEC_CYCLES" in r:
            line += "  COEXEC/BUSY %.2f" % (r["SQ_VALU_MFMA_COEXEC_CYCLES"] / r["SQ_VALU_MFMA_BUSY_CYCLES"])
        if "SQ_INSTS_VALU" in r and "SQ_INSTS_MFMA" in r:
            line += "  VALU(non-MFMA)/MFMA %.2f" % ((r["SQ_INSTS_VALU"] - r["SQ_INSTS_MFMA"]) / r["SQ_INSTS_MFMA"])
        if "FETCH_SIZE" in r:
            line += "  HBM bytes (2*FETCH+WRITE)*1024 = %.4g" % ((2 * r["FETCH_SIZE"] + r.get("WRITE_SIZE", 0)) * 1024)
        print(line)
P
