#!/bin/bash
# round-2 session 7, GPU call B: the wave-pair dK+dV pass — parity, A/B timing against the two separate passes, per-kernel trace
cd /root/repo; export TMPDIR=/tmp; OUT=gpurun_out/r05c; mkdir -p $OUT
(timeout 900 python -m pytest tests/test_backward_gpu.py -x -q 2>&1 | tail -15) > $OUT/bwd_tests.log; tail -4 $OUT/bwd_tests.log
FA2_BWD_PAIR=0 python tools/compare_sdpa.py > $OUT/compare_pair0.txt 2>&1
FA2_BWD_PAIR=1 python tools/compare_sdpa.py > $OUT/compare_pair1.txt 2>&1
echo "--- pair=0"; cat $OUT/compare_pair0.txt | cut -c1-140; echo "--- pair=1"; cat $OUT/compare_pair1.txt | cut -c1-140
bash tools/bwd_profile.sh > $OUT/bwd_kernels_pair1.txt 2>&1; cat $OUT/bwd_kernels_pair1.txt | cut -c1-200
