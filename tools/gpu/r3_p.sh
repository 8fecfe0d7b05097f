#!/bin/bash
# Round 3, pass p: step-1 statistics of the wide shape (M = 8) staged through LDS (k_cov_split_lds<8, 0>) against k_cov_split's per-wave fetches, same box.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "cov or wide or room or iterated" 2>&1 | tail -2
for v in base cov1old base cov1old; do
if [ $v = base ]; then L=disco_amd/lib/libdisco_hip.so; else L=exp_libs/libdisco_$v.so; fi
DISCO_HIP_LIB=$PWD/$L timeout 600 python bench.py --config C5 --no-cpu-baseline > gpurun_out/r03_p_C5_$v.json 2> gpurun_out/r03_p_C5_$v.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads(open('gpurun_out/r03_p_C5_$v.json').read().strip().splitlines()[-1])
print('C5 $v', round(d['ms_per_step'], 3), 'ms', 'parity', d['parity_sample']['ok'], d['parity_sample']['worst_rel'], {k: round(x['ms'], 3) for k, x in d['stages'].items()})
PY
done
