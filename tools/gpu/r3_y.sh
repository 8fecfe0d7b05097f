#!/bin/bash
# Round 3, pass y: non-temporal stores -- first the X store of the STFT kernels (DISCO_X_NT, taken), then also the mask and output-sample stores (DISCO_OUT_NT) -- C3 and C5, same box.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
for v in base outnt base outnt; do
if [ $v = base ]; then L=disco_amd/lib/libdisco_hip.so; else L=exp_libs/libdisco_$v.so; fi
DISCO_HIP_LIB=$PWD/$L timeout 600 python bench.py --extras C5 --no-cpu-baseline > gpurun_out/r03_y_$v.json 2> gpurun_out/r03_y_$v.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads(open('gpurun_out/r03_y_$v.json').read())
print('$v C3', round(d['ms_per_step'], 3), 'parity', d['parity_sample']['worst_rel_all_ranks'], {k: round(x['ms'], 3) for k, x in d['stages'].items()}, 'C5', round(d['configs']['C5']['ms_per_step'], 3), {k: round(x['ms'], 3) for k, x in d['configs']['C5']['stages'].items() if k in ('stft', 'cov1')})
PY
done
