#!/bin/bash
# Round 3, pass x: the register / DPP solver for 5 <= P <= 8 (option solve_dpp = 2: one pencil per 16-lane row, half the lanes idle) against the LDS
# group solver: full matrices, then C3 (solve2: P = 7) and C5 (solve1: P = 8) on one box.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
PYTHONPATH=. timeout 300 python tools/gpu/solve_time.py 1028000 5 7 8
for v in 1 2 1 2; do
DISCO_SOLVE_DPP=$v timeout 600 python bench.py --extras C5 --no-cpu-baseline > gpurun_out/r03_x_dpp$v.json 2> gpurun_out/r03_x_dpp$v.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads(open('gpurun_out/r03_x_dpp$v.json').read())
print('solve_dpp=$v C3', round(d['ms_per_step'], 3), 'parity', d['parity_sample']['worst_rel_all_ranks'], {k: round(x['ms'], 3) for k, x in d['stages'].items() if 'solve' in k}, 'C5', round(d['configs']['C5']['ms_per_step'], 3), d['configs']['C5']['parity_sample']['worst_rel'], {k: round(x['ms'], 3) for k, x in d['configs']['C5']['stages'].items() if 'solve' in k})
PY
done
