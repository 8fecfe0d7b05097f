#!/bin/bash
# Round 5, pass f: node-sharded driver on the fused step-1 kernel (+ half-batch overlap path), C2 as one hipGraph, the reordered plain line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "shard or graph or thread_vs_group or online" 2>&1 | tail -4
timeout 300 python bench.py --shard nodes --rooms 250 --extras none --no-cpu-baseline > gpurun_out/r5_f_bench_nodeshard.json 2>gpurun_out/r5_f_nodeshard.err; echo rc $?
python - <<PY
import json
d = json.loads(open('gpurun_out/r5_f_bench_nodeshard.json').read().strip().splitlines()[-1])
print('nodeshard 250 rooms', round(d['ms_per_step'], 3), 'ms', d['parity_sample']['worst_rel_all_ranks'], d.get('exchange', {}).get('ms_per_gather'))
PY
timeout 900 python bench.py > gpurun_out/r5_f_bench_default.json 2> gpurun_out/r5_f_bench_default.err; echo "bench rc $?"; tail -3 gpurun_out/r5_f_bench_default.err
python - <<PY
import json
d = json.loads(open('gpurun_out/r5_f_bench_default.json').read().strip().splitlines()[-1])
print(json.dumps(d['summary']))
PY
