#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r02g
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
timeout 600 python tools/kbench.py run --cfg c3,c4,c2k,c4k_b8 --rounds 5 > $O/kbench.log 2>&1; grep "median\|^--" $O/kbench.log | cut -c1-180
hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_peak.hip -o tools/ubench/mfma_peak && tools/ubench/mfma_peak > $O/mfma_peak.json; cat $O/mfma_peak.json
python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; cat $O/bench_driver.json
for wl in c2 c3 c4; do bash tools/profile_gpu.sh r02g $wl > $O/prof_$wl.log 2>&1; grep "hbm_bytes_per_launch (\|MFMA pipe busy\|fwd_d128" $O/prof_$wl.log | cut -c1-250; done
