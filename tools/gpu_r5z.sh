#!/bin/bash
# round-2 session 7, evidence pass with the final library: whole GPU suite, smoke, driver-style bench, rocprofv3 summary of c2, backward kernel trace,
# SDPA comparison, backward A/B (separate passes vs wave pairs)
set -u
export TMPDIR=/tmp
O=gpurun_out/r05z
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/gpu_tests.log; cat $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; cut -c1-1500 $O/bench_driver.json
timeout 400 python tools/compare_sdpa.py 2>&1 | grep -v amdgpu.ids > $O/compare_sdpa.txt; cat $O/compare_sdpa.txt
python tools/bwd_pair_ab.py 2>&1 | grep -v amdgpu.ids > $O/bwd_pair_ab.txt; cat $O/bwd_pair_ab.txt
bash tools/bwd_profile.sh 2>&1 | grep "fa2::" | head -12 | cut -c1-150 > $O/bwd_kernels.txt; cat $O/bwd_kernels.txt
bash tools/profile_gpu.sh r05z c2 > $O/prof_c2.log 2>&1; grep "hbm_bytes_per_launch (\|SQ_VALU_MFMA_BUSY\|GRBM_GUI" $O/prof_c2.log | cut -c1-200
