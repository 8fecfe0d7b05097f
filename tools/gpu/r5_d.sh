#!/bin/bash
# Round 5, pass d: after the pruning -- full GPU suite, then the plain bench line (summary key, stream numbers, wider parity samples)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
T0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r5_d_tests.log 2>&1; echo "tests rc $? ($(( $(date +%s) - T0 )) s)"; grep -E "passed|failed" gpurun_out/r5_d_tests.log | tail -2; grep -E "^FAILED|^E  " gpurun_out/r5_d_tests.log | head -10
T1=$(date +%s)
timeout 900 python bench.py > gpurun_out/r5_d_bench_default.json 2> gpurun_out/r5_d_bench_default.err; echo "bench rc $? ($(( $(date +%s) - T1 )) s)"; tail -3 gpurun_out/r5_d_bench_default.err
python - <<PY
import json
d = json.loads(open('gpurun_out/r5_d_bench_default.json').read().strip().splitlines()[-1])
print(json.dumps(d['summary'], indent=0))
print(json.dumps(d['configs']['online1'].get('stream'), indent=0))
print(len(json.dumps(d)), 'bytes; summary', len(json.dumps(d['summary'])))
PY
