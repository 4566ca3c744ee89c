#!/bin/bash
# after the 4-wave register budget + the 1..1.5-round rule: forward suite, row-shape probe (auto vs previous library), SDPA comparison, SD step proxy
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r05n; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) | tee $O/gputest.log
PREV=/root/repo/flash-attention-v2-rdna3-minimal_amd/libfa2_gfx950_prev.so
for rep in 1 2; do
  FA2_FRONTEND=py FA2_GFX950_LIB=$PREV python tools/rows_probe.py 2>&1 | grep -v amdgpu.ids | sed 's/^/prev  /'
  FA2_FRONTEND=py python tools/rows_probe.py 2>&1 | grep -v amdgpu.ids | sed 's/^/new   /'
done | tee $O/rows_probe_ab.txt
timeout 400 python tools/compare_sdpa.py 2>&1 | grep -v amdgpu.ids | tee $O/compare_sdpa.txt
timeout 600 python tools/sd_step_proxy.py > $O/sd_step_proxy.json 2> $O/sd_step_proxy.err; python -c "
import json; d=json.load(open('$O/sd_step_proxy.json'))
for k,v in d.items():
    if isinstance(v,dict): print(k, v.get('fa2_eager_ms'), v.get('fa2_graph_ms'), v.get('sdpa_eager_ms'), v.get('speedup_graph'))
"
