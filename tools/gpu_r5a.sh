#!/bin/bash
# round-2 session 7, GPU call A: randomised parity sweep + wall-vs-event overhead of the contractual region
cd /root/repo; export TMPDIR=/tmp; OUT=gpurun_out/r05b; mkdir -p $OUT
timeout 900 python tools/fuzz_parity.py --cases 600 --seed 1 --out $OUT/fuzz_seed1.json > $OUT/fuzz_seed1.log 2>&1
tail -3 $OUT/fuzz_seed1.log
for i in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-backward > $OUT/bench_base_$i.json 2>$OUT/bench_base_$i.err
  HSA_ENABLE_INTERRUPT=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-backward > $OUT/bench_noint_$i.json 2>$OUT/bench_noint_$i.err
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05b/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['launch_ms'], d['steady']['tflops'])
    except Exception as e: print(f, 'ERR', e)
P
