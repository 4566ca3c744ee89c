#!/bin/bash
# final round-2 evidence pass: full GPU suite, driver-style bench, default bench, rocprofv3 summaries of c2/c3/c4, backward trace
set -u
export TMPDIR=/tmp
O=gpurun_out/r02z
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; cut -c1-700 $O/bench_driver.json
for wl in c2 c3 c4; do bash tools/profile_gpu.sh r02z $wl > $O/prof_$wl.log 2>&1; grep "hbm_bytes_per_launch (\|MFMA pipe busy\|fwd_d128" $O/prof_$wl.log | cut -c1-200; done
bash tools/bwd_profile.sh > $O/bwd.log 2>&1; grep "ILi128" $O/bwd.log | head -6
