#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r03b
mkdir -p $O
for i in 1 2 3; do
  sleep 2
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cold_$i.json 2> $O/bench_cold_$i.err
  python - <<PY
import json
d=json.load(open("$O/bench_cold_$i.json"))
print("cold run $i value", d["value"], "kernel_ms", d["roofline"]["kernel_ms"], "launch", d["launch_ms"], "steady", d["steady"]["tflops"], "bwd", d["backward"]["bwd_ms"])
PY
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-backward > $O/bench_cold_nobwd.json 2> $O/bench_cold_nobwd.err; python -c "
import json;d=json.load(open('$O/bench_cold_nobwd.json'));print('no-bwd value',d['value'],d['launch_ms'],d['steady']['tflops'])"
