#!/bin/bash
# full parity suite + short bench lines of C3, C5, C2x4000; optional harness A/B
mkdir -p gpurun_out
TAG=${1:-p}
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1
tail -3 gpurun_out/${TAG}_tests.log
for c in C3 C5; do
  timeout 300 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_$c.json 2> gpurun_out/${TAG}_bench_$c.err
  python - <<P
import json
try:
    d = json.loads(open('gpurun_out/${TAG}_bench_$c.json').read().strip().splitlines()[-1])
    print('$c', round(d['ms_per_step'], 3), d['parity_sample']['worst_rel'], d['roofline']['pipeline']['frac'], {k: round(v['ms'], 3) for k, v in d.get('stages', {}).items()})
except Exception as e:
    print('$c failed', e); print(open('gpurun_out/${TAG}_bench_$c.err').read()[-2000:])
P
done
for m in 4 2; do KB_M=$m KB_GLOB="cov_bench_m${m}_*.so" python tools/gpu/kbench/run_cov_bench.py 2>&1 | grep cov_bench; done
