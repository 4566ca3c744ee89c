#!/bin/bash
# driver-style cold runs (--steps 20 --warmup 5, fresh process, idle GPU before each) of library variants
O=gpurun_out/cold
mkdir -p $O
for rep in 1 2 3; do
  for v in "$@"; do
    sleep 3
    FA2_GFX950_LIB=tools/variants/$v.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-backward 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('$v rep $rep value %.1f  kernel_ms %.4f  first %.4f last %.4f min %.4f  steady %.1f' % (r['value'], r['roofline']['kernel_ms'], r['launch_ms']['first'], r['launch_ms']['last'], r['launch_ms']['min'], r['steady']['tflops']))"
  done
done
