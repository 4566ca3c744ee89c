#!/bin/bash
# round-4 evidence pass with the final library: whole GPU suite, smoke, driver-style bench, rocprofv3 summary of c2, mask bench, SDPA comparison
set -u
export TMPDIR=/tmp
O=gpurun_out/r04z
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/gpu_tests.log; cat $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; cut -c1-1200 $O/bench_driver.json
bash tools/profile_gpu.sh r04z c2 > $O/prof_c2.log 2>&1; grep "hbm_bytes_per_launch (\|MFMA pipe busy\|fwd_d128" $O/prof_c2.log | cut -c1-200
timeout 300 python tools/mask_bench.py 2>&1 | grep -v amdgpu.ids > $O/mask_bench.txt; cat $O/mask_bench.txt
timeout 400 python tools/compare_sdpa.py 2>&1 | grep -v amdgpu.ids > $O/compare_sdpa.txt; cat $O/compare_sdpa.txt
