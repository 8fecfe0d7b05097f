#!/bin/bash
# Round 3, pass i: launch-geometry sweep of the single-node path at its own batch size (C2: 256 rooms).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
python - <<'PY'
import json, subprocess, sys
def run(tuning):
    cmd = [sys.executable, 'bench.py', '--config', 'C2', '--extras', 'none', '--no-cpu-baseline', '--no-parity', '--steps', '50', '--warmup', '5']
    if tuning: cmd += ['--tuning', tuning]
    p = subprocess.run(cmd, capture_output=True, text=True)
    for l in p.stdout.splitlines():
        if l.startswith('{'):
            d = json.loads(l)
            return d['ms_per_step'], {s: x['ms'] for s, x in d['stages'].items()}
    return None, p.stderr[-300:]
print('default', run(None))
for runw in (6, 8, 10, 14, 20, 40):
    print('runw', runw, run(f'{runw},0,0,0'))
for pairs in (4, 6, 8, 12, 16, 32):
    print('pairs', pairs, run(f'0,0,0,{pairs}'))
PY
