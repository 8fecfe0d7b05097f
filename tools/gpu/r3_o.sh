#!/bin/bash
# Round 3, pass o: full GPU suite on the DPP solver / k_apply_mq build, the plain bench line, C5 with the overlapped form forced.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 900 python bench.py > gpurun_out/r03_o_bench_default.json 2> gpurun_out/r03_o_bench_default.err; echo "bench rc $?"; tail -2 gpurun_out/r03_o_bench_default.err
python - <<PY
import json
d = json.loads(open('gpurun_out/r03_o_bench_default.json').read().strip().splitlines()[-1])
print('C3', round(d['ms_per_step'], 3), 'ms', d['roofline']['kernel'], d['roofline']['frac'], 'pipe', d['roofline']['pipeline']['frac'], 'parity', d['parity_sample']['worst_rel_all_ranks'], {k: round(x['ms'], 3) for k, x in d['stages'].items()})
for k, v in d.get('configs', {}).items():
    rf = v.get('roofline') or {}
    print(k, round(v['ms_per_step'], 3), 'ms', 'xRT', round(v['x_realtime'], 1), rf.get('kernel', '')[:30], rf.get('frac'), 'pipe', (rf.get('pipeline') or {}).get('frac'), 'ok', v['parity_sample']['ok'], {s: round(x['ms'], 2) for s, x in v['stages'].items()})
PY
for o in 2 1; do
DISCO_OVERLAP_SOLVES=$o timeout 600 python bench.py --config C5 --no-cpu-baseline > gpurun_out/r03_o_C5_overlap$o.json 2> gpurun_out/r03_o_C5_overlap$o.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads(open('gpurun_out/r03_o_C5_overlap$o.json').read().strip().splitlines()[-1])
print('C5 overlap=$o', round(d['ms_per_step'], 3), 'ms', 'pipe', d['roofline']['pipeline']['frac'], 'parity ok', d['parity_sample']['ok'])
PY
done
