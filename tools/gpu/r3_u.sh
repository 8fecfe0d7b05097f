#!/bin/bash
# Round 3, pass u: fewer squarings, more power steps (the distributed power step of the DPP solver costs 1/13 of a squaring): stop threshold 1 - tau < 0.2 / 0.5 / 0.7
# with 5 / 8 / 12 power steps -- solver parity tests first, then C3 and C5 times on one box.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
for v in sq05p8 sq07p12; do
DISCO_HIP_LIB=$PWD/exp_libs/libdisco_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "solver" 2>&1 | tail -2
done
for v in base sq05p8 sq07p12 base sq05p8 sq07p12; do
if [ $v = base ]; then L=disco_amd/lib/libdisco_hip.so; else L=exp_libs/libdisco_$v.so; fi
DISCO_HIP_LIB=$PWD/$L timeout 600 python bench.py --config C5 --no-cpu-baseline > gpurun_out/r03_u_C5_$v.json 2> gpurun_out/r03_u_C5_$v.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads(open('gpurun_out/r03_u_C5_$v.json').read().strip().splitlines()[-1])
print('C5 $v', round(d['ms_per_step'], 3), 'ms', 'parity', d['parity_sample']['ok'], d['parity_sample']['worst_rel'], {k: round(x['ms'], 3) for k, x in d['stages'].items() if 'solve' in k})
PY
done
