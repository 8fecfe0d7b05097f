#!/bin/bash
# Round 3, pass m: the wide-shape filter pass with contiguous granule loads (k_apply_mq) against k_apply_m (exp_libs/libdisco_apply32k.so), same box.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "apply or wide or room or iterated" 2>&1 | tail -2
for v in base apply32k base apply32k; do
if [ $v = base ]; then L=disco_amd/lib/libdisco_hip.so; else L=exp_libs/libdisco_$v.so; fi
DISCO_HIP_LIB=$PWD/$L timeout 600 python bench.py --config C5 --no-cpu-baseline > gpurun_out/r03_m_C5_$v.json 2> gpurun_out/r03_m_C5_$v.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads(open('gpurun_out/r03_m_C5_$v.json').read().strip().splitlines()[-1])
print('C5 $v', round(d['ms_per_step'], 3), 'ms', {k: round(x['ms'], 3) for k, x in d['stages'].items()})
PY
done
