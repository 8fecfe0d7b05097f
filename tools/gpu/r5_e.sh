#!/bin/bash
# Round 5, pass e: the online mode with its power steps on packed float32 (all but the last): step time, stage times, parity of 6 rooms, stream
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python bench.py --rooms 1000 --online-every 1 --steps 3 --warmup 1 --extras none --no-cpu-baseline > gpurun_out/r5_e_online1.json 2> gpurun_out/r5_e_online1.err; echo rc $?
python - <<PY
import json
d = json.loads(open('gpurun_out/r5_e_online1.json').read().strip().splitlines()[-1])
print('online1', round(d['ms_per_step'], 2), 'ms', round(d['x_realtime'], 1), 'x', {k: v['ms'] for k, v in d['stages'].items()}, d['parity_sample']['per_room'])
print(d.get('stream'))
PY
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "online" 2>&1 | tail -3
