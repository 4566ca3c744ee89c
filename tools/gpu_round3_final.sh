#!/bin/bash
# round-3 evidence pass: smoke, driver-style bench (f32 scale and folded scale), rocprofv3 summaries of c2/c3/c4, backward trace,
# socket power / clock samples under the steady loop
set -u
export TMPDIR=/tmp
O=gpurun_out/r03z
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; cut -c1-900 $O/bench_driver.json
FA2_D128_FOLD=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-backward > $O/bench_driver_fold.json 2> $O/bench_driver_fold.err; cut -c1-300 $O/bench_driver_fold.json
FA2_D128_PERSIST=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-backward > $O/bench_driver_nopersist.json 2> $O/bench_driver_nopersist.err; cut -c1-300 $O/bench_driver_nopersist.json
for wl in c2 c3 c4; do bash tools/profile_gpu.sh r03z $wl > $O/prof_$wl.log 2>&1; grep "hbm_bytes_per_launch (\|MFMA pipe busy\|fwd_d128" $O/prof_$wl.log | cut -c1-200; done
bash tools/bwd_profile.sh > $O/bwd.log 2>&1; grep "ILi128" $O/bwd.log | head -6
python - > $O/steady.log 2>&1 <<'PY' &
import sys, os, time, torch
sys.path.insert(0, "flash-attention-v2-rdna3-minimal_amd")
from rocwmma_fattn.FlashAttn import FlashAttentionFunction as F
q,k,v=(torch.rand((2,16,4096,128),device="cuda").half() for _ in range(3))
t0=time.time()
while time.time()-t0 < 12:
    for _ in range(200): F.apply(q,k,v,None,False)
    torch.cuda.synchronize()
PY
sleep 6
for i in 1 2 3; do rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -i "Package Power\|sclk\|junction" ; sleep 1; done > $O/power.txt
rocm-smi --showmaxpower 2>&1 | grep -i "power" >> $O/power.txt
wait
cat $O/power.txt
