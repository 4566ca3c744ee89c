#!/bin/bash
# round-2 session 7, GPU call C: full GPU suite with the wave-pair dK+dV pass, randomised sweep (backward every 2nd case), bench line, backward kernel trace
cd /root/repo; export TMPDIR=/tmp; OUT=gpurun_out/r05f; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12) > $OUT/gputest.log; tail -3 $OUT/gputest.log
timeout 900 python tools/fuzz_parity.py --cases 500 --seed 2 --bwd-every 2 --out $OUT/fuzz_seed2.json > $OUT/fuzz_seed2.log 2>&1; tail -2 $OUT/fuzz_seed2.log | cut -c1-400
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench.err; cut -c1-1200 $OUT/bench_driver_args.json
bash tools/bwd_profile.sh > $OUT/bwd_kernels.txt 2>&1; grep "bwd_" $OUT/bwd_kernels.txt | cut -c1-160 | head -8
