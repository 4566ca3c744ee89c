#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r04b
mkdir -p $O
timeout 300 python tools/diag_mask_tmp.py 2>&1 | grep -v amdgpu.ids | tee $O/diag.txt
timeout 900 python -m pytest tests/test_bias_gpu.py tests/test_parity_gpu.py -m gpu -x -q -k "bias or mask or sd_hook" 2>&1 | tail -5 | tee $O/gpu_tests.log
timeout 300 python tools/mask_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/mask_bench.txt
timeout 400 python tools/compare_sdpa.py 2>&1 | grep -v amdgpu.ids | tee $O/compare_sdpa.txt
