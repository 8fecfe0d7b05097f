#!/bin/bash
# Round 3, pass a: full GPU suite on the multi-unit build, the plain bench line (every BASELINE config), overlap A/B on C3,
# kernel stats + HBM counters + SQ/LDS counters of C3 and of the C2 shape.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3a_tests.log 2>&1; echo "tests rc $? ($(( $(date +%s) - T0 )) s)"; tail -4 gpurun_out/r3a_tests.log
T1=$(date +%s)
timeout 900 python bench.py > gpurun_out/r3a_bench_all.json 2> gpurun_out/r3a_bench_all.err; echo "bench rc $? ($(( $(date +%s) - T1 )) s)"; tail -3 gpurun_out/r3a_bench_all.err
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r3a_bench_all.json').read().strip().splitlines()[-1])
    print('C3', round(d['ms_per_step'], 3), 'ms', d['roofline']['kernel'], d['roofline']['frac'], 'pipe', d['roofline']['pipeline']['frac'], 'parity', d['parity_sample'] and d['parity_sample']['worst_rel_all_ranks'])
    for k, v in d.get('configs', {}).items():
        if 'error' in v:
            print(k, 'ERROR', v['error'][:300]); continue
        rf = v.get('roofline') or {}
        print(k, round(v['ms_per_step'], 3), 'ms', 'xRT', round(v['x_realtime'], 1), rf.get('kernel', '')[:40], rf.get('frac'), 'pipe', (rf.get('pipeline') or {}).get('frac'),
              'parity', (v.get('parity_sample') or {}).get('worst_rel_all_ranks'))
        print('   ', {s: x['ms'] for s, x in (v.get('stages') or {}).items()})
    print('   C3 stages', {s: x['ms'] for s, x in d['stages'].items()})
except Exception as e:
    print('summary failed', e)
PY
for mode in 1 0; do
  DISCO_OVERLAP_SOLVES=$mode timeout 300 python bench.py --extras none --no-cpu-baseline --no-parity --steps 20 > gpurun_out/r3a_bench_C3_overlap$mode.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/r3a_bench_C3_overlap$mode.json').read().strip().splitlines()[-1]); print('overlap=$mode', round(d['ms_per_step'],3), {s:x['ms'] for s,x in d['stages'].items()})"
done
bash tools/profile_round.sh r03_a_C3 2>&1 | tail -14
BARGS="" bash tools/gpu/pmc_alu.sh r03_a_C3 2>&1 | tail -20
BARGS="--config C2 --rooms 4000" bash tools/gpu/pmc_alu.sh r03_a_C2x4000 2>&1 | tail -12
echo "total $(( $(date +%s) - T0 )) s"
