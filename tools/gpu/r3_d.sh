#!/bin/bash
# Round 3, pass d: online tracking solve A/B (default = tracking, 3 waves/SIMD; wpe2; track0 = round 2's full solve per frame),
# GPU suite, full line, C3 / C5 kernel stats + HBM counters for profiles/.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3d_tests.log 2>&1; echo "tests rc $? ($(( $(date +%s) - T0 )) s)"; grep -E "passed|failed" gpurun_out/r3d_tests.log | tail -2; grep -E "^FAILED|^E  " gpurun_out/r3d_tests.log | head -10
for lib in default wpe2 track0; do
  if [ $lib = default ]; then L=""; else L="DISCO_HIP_LIB=$GRAFT_REPO_ROOT/exp_libs/libdisco_$lib.so"; fi
  env $L timeout 600 python bench.py --rooms 1000 --online-every 1 --steps 2 --warmup 1 --extras none --no-cpu-baseline > gpurun_out/r3d_online_$lib.json 2>gpurun_out/r3d_err.log || tail -3 gpurun_out/r3d_err.log
  python -c "
import json; d=json.loads(open('gpurun_out/r3d_online_$lib.json').read().strip().splitlines()[-1]); print('online1 $lib', round(d['ms_per_step'],2), 'ms xRT', round(d['x_realtime'],1), {s:x['ms'] for s,x in d['stages'].items()}, 'parity', d['parity_sample']['per_room'])"
done
env timeout 600 python bench.py --rooms 1000 --online-every 8 --steps 2 --warmup 1 --extras none --no-cpu-baseline > gpurun_out/r3d_online8.json 2>gpurun_out/r3d_err.log || tail -3 gpurun_out/r3d_err.log
python -c "
import json; d=json.loads(open('gpurun_out/r3d_online8.json').read().strip().splitlines()[-1]); print('online8', round(d['ms_per_step'],2), 'ms xRT', round(d['x_realtime'],1), {s:x['ms'] for s,x in d['stages'].items()}, 'parity', d['parity_sample']['per_room'])"
T1=$(date +%s)
timeout 900 python bench.py > gpurun_out/r3d_bench_all.json 2> gpurun_out/r3d_bench_all.err; echo "bench rc $? ($(( $(date +%s) - T1 )) s)"; tail -3 gpurun_out/r3d_bench_all.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3d_bench_all.json').read().strip().splitlines()[-1])
print('C3', round(d['ms_per_step'], 3), 'ms', d['roofline']['kernel'], d['roofline']['frac'], 'pipe', d['roofline']['pipeline']['frac'], 'parity', d['parity_sample'] and d['parity_sample']['worst_rel_all_ranks'])
print('   ', {s: x['ms'] for s, x in d['stages'].items()})
for k, v in d.get('configs', {}).items():
    if 'error' in v:
        print(k, 'ERROR', v['error'][:300]); continue
    rf = v.get('roofline') or {}
    print(k, round(v['ms_per_step'], 3), 'ms', 'xRT', round(v['x_realtime'], 1), rf.get('kernel', '')[:40], rf.get('frac'), 'pipe', (rf.get('pipeline') or {}).get('frac'), 'parity', (v.get('parity_sample') or {}).get('per_room'), 'ok', (v.get('parity_sample') or {}).get('ok'))
PY
bash tools/profile_round.sh r03_d_C3 2>&1 | tail -9
bash tools/profile_round.sh r03_d_C5 --config C5 2>&1 | tail -12
echo "total $(( $(date +%s) - T0 )) s"
