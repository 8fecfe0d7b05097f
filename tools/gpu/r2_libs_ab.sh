#!/bin/bash
# A/B of whole-library builds under exp_libs/ on one config: usage r2_libs_ab.sh TAG CONFIG lib1 lib2 ...
mkdir -p gpurun_out
TAG=$1; CFG=$2; shift 2
cp disco_amd/lib/libdisco_hip.so /tmp/lib_keep.so
for l in "$@"; do
  cp exp_libs/lib_$l.so disco_amd/lib/libdisco_hip.so
  for rep in 1 2; do
    timeout 300 python bench.py --config $CFG --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_${CFG}_$l.json 2> gpurun_out/${TAG}_${CFG}_$l.err
    python - <<P
import json
try:
    d = json.loads(open('gpurun_out/${TAG}_${CFG}_$l.json').read().strip().splitlines()[-1])
    print('$CFG $l', round(d['ms_per_step'], 3), d['parity_sample']['worst_rel'], {k: round(v['ms'], 3) for k, v in d.get('stages', {}).items()})
except Exception as e:
    print('$CFG $l failed', e); print(open('gpurun_out/${TAG}_${CFG}_$l.err').read()[-1500:])
P
  done
done
cp /tmp/lib_keep.so disco_amd/lib/libdisco_hip.so
