#!/bin/bash
# PMC counters of the backward kernels at the config-2 shape (developer tool; separate --pmc passes, kernel trace only)
cd /root/repo; export TMPDIR=/tmp; OUT=gpurun_out/bwdpmc; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/bwd_c2.py <<'P'
import os, sys, torch
sys.path.insert(0, "/root/repo/flash-attention-v2-rdna3-minimal_amd")
from rocwmma_fattn.FlashAttn import FlashAttentionFunction
g = torch.Generator(device="cuda").manual_seed(1)
q, k, v = (torch.rand((2, 16, 4096, 128), generator=g, device="cuda").half().requires_grad_(True) for _ in range(3))
o = FlashAttentionFunction.apply(q, k, v, None, False)
go = torch.rand(o.shape, generator=g, device="cuda").half()
for _ in range(6):
    o.backward(go, retain_graph=True)
torch.cuda.synchronize()
P
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES" \
           "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  FA2_BWD_PAIR=${FA2_BWD_PAIR:-1} rocprofv3 --kernel-trace --output-format csv --pmc $set -d $OUT/p$i -o pmc -- python /tmp/bwd_c2.py > $OUT/p$i.log 2>&1
done
python - <<'P'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for path in glob.glob("gpurun_out/bwdpmc/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        kn = r.get("Kernel_Name", "")
        if "bwd_" not in kn: continue
        key = (kn.split("(")[0].replace("void fa2::", ""), r["Counter_Name"])
        acc[key][0] += float(r["Counter_Value"] or 0); acc[key][1] += 1
for (kn, c), (s, n) in sorted(acc.items()):
    print("%-50s %-28s %14.4g  (%d launches)" % (kn, c, s / n, n))
P
