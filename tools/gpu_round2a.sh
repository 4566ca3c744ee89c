#!/bin/bash
# round-2 first GPU pass: test suite, MFMA peak, soffset A/B + poison test, driver-style bench, c4/c2 profiles, backward trace
set -u
export TMPDIR=/tmp
O=gpurun_out/r02a
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_peak.hip -o tools/ubench/mfma_peak && tools/ubench/mfma_peak > $O/mfma_peak.json; cat $O/mfma_peak.json
FA2_GFX950_LIB=tools/variants/soff.so python -m pytest tests/test_parity_gpu.py -m gpu -q -k "ragged_tail or seeded" > $O/pytest_soff.log 2>&1; tail -3 $O/pytest_soff.log
python tools/kbench.py run --cfg c2,c3,c4 --rounds 7 base soff > $O/kbench.log 2>&1; tail -12 $O/kbench.log
python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; cat $O/bench_driver.json
bash tools/profile_gpu.sh r02a c4 > $O/prof_c4.log 2>&1; tail -30 $O/prof_c4.log
bash tools/bwd_profile.sh > $O/bwd.log 2>&1; tail -8 $O/bwd.log
