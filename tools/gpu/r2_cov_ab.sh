#!/bin/bash
# A/B of the LDS-staged covariance on C5 + parity tests of the P > 8 shapes
mkdir -p gpurun_out
TAG=${1:-cov}
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "solver or cov_solve_apply or iterated" > gpurun_out/${TAG}_tests.log 2>&1
tail -3 gpurun_out/${TAG}_tests.log
for v in 1 0; do
  DISCO_COV_LDS=$v timeout 300 python bench.py --config C5 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_C5_lds$v.json 2> gpurun_out/${TAG}_bench_C5_lds$v.err
  python - <<P
import json
try:
    d = json.loads(open('gpurun_out/${TAG}_bench_C5_lds$v.json').read().strip().splitlines()[-1])
    print('C5 lds=$v', d['ms_per_step'], d['parity_sample']['worst_rel'], {k: round(v['ms'], 3) for k, v in d.get('stages', {}).items()})
except Exception as e:
    print('failed', e); print(open('gpurun_out/${TAG}_bench_C5_lds$v.err').read()[-2000:])
P
done
