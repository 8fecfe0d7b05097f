#!/bin/bash
# Round 3, pass l: frame chunks of the wide-shape filter pass (k_apply_m): 8 192 / 32 768 / 131 072 workgroups aimed at, same box.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
for v in base apply32k apply128k base apply32k apply128k; do
if [ $v = base ]; then L=disco_amd/lib/libdisco_hip.so; else L=exp_libs/libdisco_$v.so; fi
DISCO_HIP_LIB=$PWD/$L timeout 600 python bench.py --config C5 --no-cpu-baseline > gpurun_out/r03_l_C5_$v.json 2> gpurun_out/r03_l_C5_$v.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads(open('gpurun_out/r03_l_C5_$v.json').read().strip().splitlines()[-1])
print('C5 $v', round(d['ms_per_step'], 3), 'ms', 'parity ok', d['parity_sample']['ok'], {k: round(x['ms'], 3) for k, x in d['stages'].items()})
PY
done
