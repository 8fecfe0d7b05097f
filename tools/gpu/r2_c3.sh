#!/bin/bash
# C3 (x2) and C5 bench lines + the parity tests that touch the STFT kernels
mkdir -p gpurun_out
TAG=${1:-c3}
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "stft or end_to_end or geometry or large_batch" > gpurun_out/${TAG}_tests.log 2>&1
tail -2 gpurun_out/${TAG}_tests.log
for c in C3 C3 C5; do
  timeout 300 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_$c.json 2> gpurun_out/${TAG}_bench_$c.err
  python - <<P
import json
try:
    d = json.loads(open('gpurun_out/${TAG}_bench_$c.json').read().strip().splitlines()[-1])
    print('$c', round(d['ms_per_step'], 3), d['parity_sample']['worst_rel'], d['roofline']['frac'], d['roofline']['pipeline']['frac'], {k: round(v['ms'], 3) for k, v in d.get('stages', {}).items()})
except Exception as e:
    print('$c failed', e); print(open('gpurun_out/${TAG}_bench_$c.err').read()[-2000:])
P
done
