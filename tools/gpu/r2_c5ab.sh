#!/bin/bash
# C5 A/B of library variants: usage r2_c5ab.sh TAG name ...   (exp_libs/libdisco_<name>.so)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r02_c5ab}; shift
for l in "$@"; do
  DISCO_HIP_LIB=$PWD/exp_libs/libdisco_$l.so timeout 300 python bench.py --config C5 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_C5_$l.json 2> gpurun_out/${TAG}_C5_$l.err
  python - <<P
import json
try:
    d = json.loads([l for l in open('gpurun_out/${TAG}_C5_$l.json') if l.startswith('{')][-1])
    print('C5 $l', round(d['ms_per_step'], 3), 'parity', d['parity_sample'] and d['parity_sample']['worst_rel'], {k: round(v['ms'], 3) for k, v in (d.get('stages') or {}).items()})
except Exception as e:
    print('C5 $l failed', e); print(open('gpurun_out/${TAG}_C5_$l.err').read()[-1200:])
P
done
