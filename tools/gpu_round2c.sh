#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r02f
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -8 $O/pytest.log
timeout 600 python tools/kbench.py run --cfg c2,c3,c4,n2k,b8 --rounds 5 > $O/kbench.log 2>&1; grep "check\|per-wave\|median\|^--" $O/kbench.log | cut -c1-180
