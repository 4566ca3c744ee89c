#!/bin/bash
# round-4 first pass: whole GPU suite with the bias kernels in the library, refreshed SDPA comparison (compiled front end), mask bench
set -u
export TMPDIR=/tmp
O=gpurun_out/r04a
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/gpu_tests.log; cat $O/gpu_tests.log
timeout 300 python tools/mask_bench.py > $O/mask_bench.txt 2>&1; cat $O/mask_bench.txt
timeout 400 python tools/compare_sdpa.py > $O/compare_sdpa.txt 2>&1; cat $O/compare_sdpa.txt
