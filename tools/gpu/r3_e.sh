#!/bin/bash
# Round 3, pass e: fused STFT + step-1 covariance for the wide shape (C5) A/B, C4 with bf16 operands, GPU suite.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3e_tests.log 2>&1; echo "tests rc $? ($(( $(date +%s) - T0 )) s)"; grep -E "passed|failed" gpurun_out/r3e_tests.log | tail -2; grep -E "^FAILED|^E  " gpurun_out/r3e_tests.log | head -10
for mode in 1 0; do
  DISCO_WIDE_STFT_COV=$mode timeout 600 python bench.py --config C5 --steps 10 --extras none --no-cpu-baseline > gpurun_out/r3e_C5_wide$mode.json 2>gpurun_out/r3e_err.log || tail -3 gpurun_out/r3e_err.log
  python -c "
import json; d=json.loads(open('gpurun_out/r3e_C5_wide$mode.json').read().strip().splitlines()[-1]); print('C5 wide_stft_cov=$mode', round(d['ms_per_step'],2), 'ms pipe', d['roofline']['pipeline']['frac'], {s:x['ms'] for s,x in d['stages'].items()}, 'parity', d['parity_sample']['per_room'])"
done
timeout 900 python bench.py --extras C4,C4_bf16 --no-cpu-baseline > gpurun_out/r3e_c4.json 2> gpurun_out/r3e_c4.err; echo "c4 rc $?"; tail -3 gpurun_out/r3e_c4.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3e_c4.json').read().strip().splitlines()[-1])
for k, v in d.get('configs', {}).items():
    if 'error' in v:
        print(k, 'ERROR', v['error'][:300], v.get('trace', '')[-600:]); continue
    rf = v.get('roofline') or {}
    print(k, round(v['ms_per_step'], 3), 'ms', 'xRT', round(v['x_realtime'], 1), rf.get('kernel', '')[:60], rf.get('frac'), 'parity', (v.get('parity_sample') or {}).get('per_room'), v.get('mask_error'))
    print('    ', {s: x['ms'] for s, x in (v.get('stages') or {}).items()})
PY
echo "total $(( $(date +%s) - T0 )) s"
