#!/bin/bash
# Round 3, pass r: frame chunks of k_apply_mq (8 192 / 32 768 / 131 072 workgroups aimed at) -- time and HBM bytes, same box.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
for v in apply128k apply256k apply512k apply128k apply256k apply512k; do
if [ $v = base ]; then L=disco_amd/lib/libdisco_hip.so; else L=exp_libs/libdisco_$v.so; fi
DISCO_HIP_LIB=$PWD/$L timeout 600 python bench.py --config C5 --no-cpu-baseline --no-parity --extras none > gpurun_out/r03_r_C5_$v.json 2> gpurun_out/r03_r_C5_$v.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads(open('gpurun_out/r03_r_C5_$v.json').read().strip().splitlines()[-1])
print('C5 $v', round(d['ms_per_step'], 3), 'ms', {k: round(x['ms'], 3) for k, x in d['stages'].items()})
PY
done
