#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r03a
mkdir -p $O
timeout 900 python tools/kbench.py run --cfg c2 --rounds 7 > $O/kbench.log 2>&1; grep "median\|^--\|check\|per-wave" $O/kbench.log | cut -c1-220
for i in 1 2 3; do
  sleep 2
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-backward > $O/bench_cold_$i.json 2> $O/bench_cold_$i.err
  python - <<PY
import json
d=json.load(open("$O/bench_cold_$i.json"))
print("cold run $i value", d["value"], "launch", d["launch_ms"], "steady", d["steady"]["tflops"], "gate", d["check"])
PY
done
