#!/bin/bash
# final evidence pass of the session with the final library
set -u
export TMPDIR=/tmp
O=gpurun_out/r05z2
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/gpu_tests.log; cat $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; cut -c1-300 $O/bench_driver.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
bash tools/bwd_profile.sh 2>&1 | grep "fa2::" | head -14 | cut -c1-170 > $O/bwd_kernels.txt; head -4 $O/bwd_kernels.txt
python tools/bwd_pair_ab.py 2>&1 | grep -v amdgpu.ids > $O/bwd_pair_ab.txt; cat $O/bwd_pair_ab.txt
