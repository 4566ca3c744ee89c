#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r03c
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; cut -c1-1200 $O/bench_driver.json
FA2_D128_FOLD=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-backward > $O/bench_driver_fold.json 2> $O/bench_driver_fold.err; cut -c1-700 $O/bench_driver_fold.json
