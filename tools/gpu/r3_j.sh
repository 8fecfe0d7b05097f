#!/bin/bash
# Round 3, pass j: the plain bench line on the final bench.py (twice: box noise), the bench subprocess tests.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2; do
timeout 900 python bench.py > gpurun_out/r03_j_bench_default_$i.json 2> gpurun_out/r03_j_bench_default_$i.err; echo "bench rc $?"; tail -2 gpurun_out/r03_j_bench_default_$i.err
python - <<PY
import json
d = json.loads(open('gpurun_out/r03_j_bench_default_$i.json').read().strip().splitlines()[-1])
print('C3', round(d['ms_per_step'], 3), 'ms', d['roofline']['kernel'], d['roofline']['frac'], 'pipe', d['roofline']['pipeline']['frac'], 'parity', d['parity_sample']['worst_rel_all_ranks'], d['roofline'].get('traffic_note'))
for k, v in d.get('configs', {}).items():
    rf = v.get('roofline') or {}
    print(k, round(v['ms_per_step'], 3), 'ms', 'xRT', round(v['x_realtime'], 1), rf.get('kernel', '')[:30], rf.get('frac'), 'pipe', (rf.get('pipeline') or {}).get('frac'), 'sum stages', round(sum(x['ms'] for x in v['stages'].values()), 3), 'ok', v['parity_sample']['ok'])
PY
done
timeout 1500 python -m pytest tests/test_gpu_zz_node_sharded_torch.py -x -q 2>&1 | tail -3
