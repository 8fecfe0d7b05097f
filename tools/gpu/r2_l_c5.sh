#!/bin/bash
# C5 bench line (+ the wide-shape parity tests) after the XCD-aware workgroup ids of k_apply_m
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r02_l}
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "room_cov or iterated or end_to_end" > gpurun_out/${TAG}_tests_c5.log 2>&1; echo "tests rc $?"
tail -3 gpurun_out/${TAG}_tests_c5.log
timeout 400 python bench.py --config C5 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_quick_C5.json 2> gpurun_out/${TAG}_quick_C5.err
python - <<P
import json
d = json.loads([l for l in open('gpurun_out/${TAG}_quick_C5.json') if l.startswith('{')][-1])
print('C5 ms/step', d['ms_per_step'], d['parity_sample'] and d['parity_sample']['worst_rel'], {k: v['ms'] for k, v in d['stages'].items()})
P
