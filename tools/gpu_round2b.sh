#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r02b
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -15 $O/pytest.log
timeout 600 python tools/kbench.py run asm hip --cfg c2,c3,c4,n2k,n1k,b8 --rounds 7 > $O/kbench.log 2>&1; tail -30 $O/kbench.log
