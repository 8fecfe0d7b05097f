#!/bin/bash
# Round 4, pass q2: packed-float32 squarings with the sums advancing together; solve times + bench line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/gpu/solve_thread_time.py 1028000 4 5 6 7 8 > gpurun_out/r04_q2_solve_thread.txt 2>&1; tail -5 gpurun_out/r04_q2_solve_thread.txt
timeout 900 python bench.py --steps 10 --extras C5,online1 --no-cpu-baseline > gpurun_out/r04_q2_bench.json 2> gpurun_out/r04_q2_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04_q2_bench.json').read().strip().splitlines()[-1])
print('C3', round(d['ms_per_step'], 3), {s: x['ms'] for s, x in d['stages'].items()}, 'parity', d['parity_sample']['worst_rel_all_ranks'], d['roofline'].get('pipeline'))
for k, v in d['configs'].items():
    print('  ', k, round(v['ms_per_step'], 2), 'xRT', round(v['x_realtime'], 1), {s: x['ms'] for s, x in v['stages'].items()}, 'parity', v['parity_sample']['worst_rel_all_ranks'], v['roofline'].get('pipeline'))
PY
