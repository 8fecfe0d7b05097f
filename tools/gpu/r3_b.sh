#!/bin/bash
# Round 3, pass b: pipelined overlap (solves on the side stream) A/B on C3 and C5, the C5 room whose error exceeded 1e-4, the full line.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3b_tests.log 2>&1; echo "tests rc $? ($(( $(date +%s) - T0 )) s)"; grep -E "passed|failed" gpurun_out/r3b_tests.log | tail -2; grep -E "^FAILED|Error" gpurun_out/r3b_tests.log | head -10
for cfg in C3 C5; do
for mode in 1 0; do
  DISCO_OVERLAP_SOLVES=$mode timeout 300 python bench.py --config $cfg --extras none --no-cpu-baseline --no-parity --steps 20 > gpurun_out/r3b_bench_${cfg}_overlap$mode.json 2>gpurun_out/r3b_err.log || tail -5 gpurun_out/r3b_err.log
  python -c "
import json; d=json.loads(open('gpurun_out/r3b_bench_${cfg}_overlap$mode.json').read().strip().splitlines()[-1]); print('$cfg overlap=$mode', round(d['ms_per_step'],3), d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['pipeline']['frac'], {s:(x['ms'], x['launches_per_step']) for s,x in d['stages'].items()})"
done; done
timeout 900 python tools/gpu/dbg_c5_room.py 199,0 2>&1 | tail -12
T1=$(date +%s)
timeout 900 python bench.py > gpurun_out/r3b_bench_all.json 2> gpurun_out/r3b_bench_all.err; echo "bench rc $? ($(( $(date +%s) - T1 )) s)"; tail -3 gpurun_out/r3b_bench_all.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3b_bench_all.json').read().strip().splitlines()[-1])
print('C3', round(d['ms_per_step'], 3), 'ms', d['roofline']['kernel'], d['roofline']['frac'], 'pipe', d['roofline']['pipeline']['frac'], 'parity', d['parity_sample'] and d['parity_sample']['worst_rel_all_ranks'])
for k, v in d.get('configs', {}).items():
    if 'error' in v:
        print(k, 'ERROR', v['error'][:300]); continue
    rf = v.get('roofline') or {}
    print(k, round(v['ms_per_step'], 3), 'ms', 'xRT', round(v['x_realtime'], 1), rf.get('kernel', '')[:40], rf.get('frac'), 'pipe', (rf.get('pipeline') or {}).get('frac'), 'parity', (v.get('parity_sample') or {}).get('per_room'))
PY
echo "total $(( $(date +%s) - T0 )) s"
