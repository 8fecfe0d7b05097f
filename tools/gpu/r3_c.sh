#!/bin/bash
# Round 3, pass c: mixed-precision solver (float32 squarings + float64 Rayleigh-quotient finish) A/B, overlap modes on C3 / C5,
# covariance chunk count vs C5 accuracy, the full line.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3c_tests.log 2>&1; echo "tests rc $? ($(( $(date +%s) - T0 )) s)"; grep -E "passed|failed" gpurun_out/r3c_tests.log | tail -2; grep -E "^FAILED|^E  " gpurun_out/r3c_tests.log | head -10
run() {  # name, env..., -- bench args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py "$@" --extras none --no-cpu-baseline > gpurun_out/r3c_$name.json 2>gpurun_out/r3c_err.log || tail -3 gpurun_out/r3c_err.log
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r3c_$name.json').read().strip().splitlines()[-1])
    ps = d.get('parity_sample') or {}
    print('$name', round(d['ms_per_step'], 3), 'ms', d['roofline']['kernel'], d['roofline']['frac'], 'pipe', d['roofline']['pipeline']['frac'], 'parity', ps.get('per_room'))
    print('      ', {s: (x['ms'], x['launches_per_step']) for s, x in d['stages'].items()})
except Exception as e:
    print('$name failed', e)
PY
}
run C3_plain_f64   DISCO_OVERLAP_SOLVES=0 DISCO_SOLVE_F32=0 -- --config C3 --steps 10
run C3_plain_mixed DISCO_OVERLAP_SOLVES=0 DISCO_SOLVE_F32=1 -- --config C3 --steps 10
run C3_over1_mixed DISCO_OVERLAP_SOLVES=1 DISCO_SOLVE_F32=1 -- --config C3 --steps 20
run C5_plain_f64   DISCO_OVERLAP_SOLVES=0 DISCO_SOLVE_F32=0 -- --config C5 --steps 10
run C5_plain_mixed DISCO_OVERLAP_SOLVES=0 DISCO_SOLVE_F32=1 -- --config C5 --steps 10
run C5_over1_mixed DISCO_OVERLAP_SOLVES=1 DISCO_SOLVE_F32=1 -- --config C5 --steps 10
run C5_over3_mixed DISCO_OVERLAP_SOLVES=3 DISCO_SOLVE_F32=1 -- --config C5 --steps 10
run C5_chunks4     DISCO_OVERLAP_SOLVES=0 DISCO_SOLVE_F32=1 -- --config C5 --steps 5 --tuning 0,4,0,0
run C5_chunks8     DISCO_OVERLAP_SOLVES=0 DISCO_SOLVE_F32=1 -- --config C5 --steps 5 --tuning 0,8,0,0
T1=$(date +%s)
timeout 900 python bench.py > gpurun_out/r3c_bench_all.json 2> gpurun_out/r3c_bench_all.err; echo "bench rc $? ($(( $(date +%s) - T1 )) s)"; tail -3 gpurun_out/r3c_bench_all.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3c_bench_all.json').read().strip().splitlines()[-1])
print('C3', round(d['ms_per_step'], 3), 'ms', d['roofline']['kernel'], d['roofline']['frac'], 'pipe', d['roofline']['pipeline']['frac'], 'parity', d['parity_sample'] and d['parity_sample']['worst_rel_all_ranks'])
for k, v in d.get('configs', {}).items():
    if 'error' in v:
        print(k, 'ERROR', v['error'][:300]); continue
    rf = v.get('roofline') or {}
    print(k, round(v['ms_per_step'], 3), 'ms', 'xRT', round(v['x_realtime'], 1), rf.get('kernel', '')[:40], rf.get('frac'), 'pipe', (rf.get('pipeline') or {}).get('frac'), 'parity', (v.get('parity_sample') or {}).get('per_room'))
PY
echo "total $(( $(date +%s) - T0 )) s"
