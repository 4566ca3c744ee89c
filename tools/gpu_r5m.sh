#!/bin/bash
# A/B of the 4-wave forward kernels' register budget (previous library vs the new build), small / awkward grids
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r05m; mkdir -p $O
PREV=/root/repo/flash-attention-v2-rdna3-minimal_amd/libfa2_gfx950_prev.so
for rep in 1 2; do
  FA2_FRONTEND=py FA2_GFX950_LIB=$PREV python tools/rows_probe.py 2>&1 | grep -v amdgpu.ids | sed 's/^/prev  /'
  FA2_FRONTEND=py python tools/rows_probe.py 2>&1 | grep -v amdgpu.ids | sed 's/^/new   /'
  FA2_FRONTEND=py FA2_FWD_ROWS=128 FA2_GFX950_LIB=$PREV python tools/rows_probe.py 2>&1 | grep -v amdgpu.ids | sed 's/^/prev128 /'
  FA2_FRONTEND=py FA2_FWD_ROWS=128 python tools/rows_probe.py 2>&1 | grep -v amdgpu.ids | sed 's/^/new128  /'
done | tee $O/rows_probe_ab.txt
