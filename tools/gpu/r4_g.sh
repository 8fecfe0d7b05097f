#!/bin/bash
# Round 4, pass g: one thread per pencil for P <= 7 (batch solves and the online mode) against the LDS group solver: C3 + online lines.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
for th in 1 0; do
  DISCO_SOLVE_THREAD=$th timeout 600 python bench.py --steps 10 --extras online1 --no-cpu-baseline > gpurun_out/r04_g_bench_thread$th.json 2> gpurun_out/r04_g_bench_thread$th.err; echo "thread=$th rc $?"
done
timeout 300 python -m pytest tests -m gpu -q -x -k "solver or online" > gpurun_out/r04_g_tests.log 2>&1; echo "tests rc $?"; tail -2 gpurun_out/r04_g_tests.log
python - <<'PY'
import json
for th in (1, 0):
    d = json.loads(open(f'gpurun_out/r04_g_bench_thread{th}.json').read().strip().splitlines()[-1])
    print('thread', th, 'C3', round(d['ms_per_step'], 3), {s: x['ms'] for s, x in d['stages'].items()}, 'parity', d['parity_sample']['worst_rel_all_ranks'])
    v = d['configs']['online1']
    print('   online1', round(v['ms_per_step'], 2), 'xRT', round(v['x_realtime'], 1), {s: x['ms'] for s, x in v['stages'].items()}, 'parity', v['parity_sample']['worst_rel_all_ranks'])
PY
