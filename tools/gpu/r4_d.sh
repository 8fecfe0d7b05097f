#!/bin/bash
# Round 4, pass d: the new defaults (room_sub 8, float64 step-1 statistics with one frame chunk, z stored by the last pass only): full GPU suite
# incl. the full-length C5 test, C5 variants on nine rooms, the plain bench line.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04_d_tests.log 2>&1; echo "tests rc $? ($(( $(date +%s) - T0 )) s)"; grep -E "passed|failed" gpurun_out/r04_d_tests.log | tail -2; grep -E "^FAILED|^E  " gpurun_out/r04_d_tests.log | head -10
T1=$(date +%s)
timeout 1500 python tools/gpu/exp_c5_variants.py gpurun_out/r04_d_c5_variants.json sample=0,25,50,75,100,125,150,175,199 variants=8:64:0:0,8:64:0:0,4:64:0:0,8:8:0:0,8:64:2:0 > gpurun_out/r04_d_c5_variants.log 2>&1; echo "variants rc $? ($(( $(date +%s) - T1 )) s)"; tail -5 gpurun_out/r04_d_c5_variants.log | cut -c1-330; head -6 gpurun_out/r04_d_c5_variants.log | cut -c1-330
T2=$(date +%s)
timeout 900 python bench.py > gpurun_out/r04_d_bench_default.json 2> gpurun_out/r04_d_bench_default.err; echo "bench rc $? ($(( $(date +%s) - T2 )) s)"; tail -3 gpurun_out/r04_d_bench_default.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04_d_bench_default.json').read().strip().splitlines()[-1])
print('C3', round(d['ms_per_step'], 3), 'ms', d['roofline']['kernel'], d['roofline']['frac'], 'pipe', d['roofline']['pipeline']['frac'], 'parity', d['parity_sample'] and d['parity_sample']['worst_rel_all_ranks'])
print('   ', {s: x['ms'] for s, x in d['stages'].items()})
for k, v in d.get('configs', {}).items():
    if 'error' in v:
        print(k, 'ERROR', v['error'][:300]); continue
    rf = v.get('roofline') or {}
    print(k, round(v['ms_per_step'], 3), 'ms', 'xRT', round(v['x_realtime'], 1), rf.get('kernel', '')[:40], rf.get('frac'), 'pipe', (rf.get('pipeline') or {}).get('frac'), 'parity', (v.get('parity_sample') or {}).get('worst_rel_all_ranks'), 'ok', (v.get('parity_sample') or {}).get('ok'), rf.get('sanity_errors'))
    print('   ', {s: x['ms'] for s, x in (v.get('stages') or {}).items()})
PY
echo "total $(( $(date +%s) - T0 )) s"
