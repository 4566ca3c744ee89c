#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03d; mkdir -p $O
for f in rand zeros randn; do echo "== fill $f"; python tools/kbench.py run --cfg c2 --rounds 5 --fill $f 2>&1 | grep "median\|per-wave" | cut -c1-330; done
# power / clock sampling during a steady loop
python - > $O/steady.log 2>&1 <<'PY' &
import sys, os, time, torch
sys.path.insert(0, "flash-attention-v2-rdna3-minimal_amd")
from rocwmma_fattn.FlashAttn import FlashAttentionFunction as F
q,k,v=(torch.rand((2,16,4096,128),device="cuda").half() for _ in range(3))
t0=time.time()
while time.time()-t0 < 14:
    for _ in range(200): F.apply(q,k,v,None,False)
    torch.cuda.synchronize()
print("done")
PY
sleep 5
for i in 1 2 3; do rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -i "power\|sclk\|mclk\|fclk\|Temperature (Sensor junction)" | head -8; sleep 1.5; done
amd-smi metric -p -c 2>&1 | head -40
wait
rocm-smi --showmaxpower --showpower 2>&1 | grep -i "power" | head
