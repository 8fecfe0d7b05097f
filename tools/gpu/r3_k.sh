#!/bin/bash
# Round 3, pass k: the register / DPP solver for 9 <= P <= 16 -- GPU parity tests of the solver and the wide shapes, C5 with both solvers on one box.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "solver or wide or c5 or iterated or room or degenerate or singular" 2>&1 | tail -4
for dpp in 1 0 1; do
DISCO_SOLVE_DPP=$dpp timeout 600 python bench.py --config C5 > gpurun_out/r03_k_C5_dpp$dpp.json 2> gpurun_out/r03_k_C5_dpp$dpp.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads(open('gpurun_out/r03_k_C5_dpp$dpp.json').read().strip().splitlines()[-1])
print('C5 dpp=$dpp', round(d['ms_per_step'], 3), 'ms', 'pipe', d['roofline']['pipeline']['frac'], 'parity', d['parity_sample'], {k: round(v['ms'], 3) for k, v in d['stages'].items()})
PY
done
DISCO_OVERLAP_SOLVES=2 timeout 600 python bench.py --config C5 > gpurun_out/r03_k_C5_overlap2.json 2> gpurun_out/r03_k_C5_overlap2.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads(open('gpurun_out/r03_k_C5_overlap2.json').read().strip().splitlines()[-1])
print('C5 dpp=1 overlap=2', round(d['ms_per_step'], 3), 'ms', 'parity ok', d['parity_sample']['ok'])
PY
PYTHONPATH=. timeout 300 python tools/gpu/solve_time.py 820800 15 9 12 16
