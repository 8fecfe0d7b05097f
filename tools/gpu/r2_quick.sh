#!/bin/bash
# quick pass: solver / covariance parity tests, then short bench lines of C3, C5, C2x4000 and the online mode
mkdir -p gpurun_out
TAG=${1:-q}
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "solver or cov_solve_apply or iterated or online" > gpurun_out/${TAG}_tests.log 2>&1
tail -3 gpurun_out/${TAG}_tests.log
for c in C3 C5; do
  timeout 300 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_$c.json 2> gpurun_out/${TAG}_bench_$c.err
  python - <<P
import json
try:
    d = json.loads(open('gpurun_out/${TAG}_bench_$c.json').read().strip().splitlines()[-1])
    print('$c', d['ms_per_step'], d['value'], d['parity_sample'], {k: round(v['ms'], 3) for k, v in d.get('stages', {}).items()})
except Exception as e:
    print('$c failed', e); print(open('gpurun_out/${TAG}_bench_$c.err').read()[-2000:])
P
done
timeout 300 python bench.py --rooms 200 --online-every 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_online1.json 2>&1
tail -c 600 gpurun_out/${TAG}_bench_online1.json
