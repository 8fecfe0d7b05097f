#!/bin/bash
# full -m gpu suite on the in-tree library, bench lines of C3 / C2 / C2x4000 / C5, and library variants on C5 (exp_libs/libdisco_<name>.so)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r02_j}; shift
timeout 1500 python -m pytest tests -m gpu -q -x -rs > gpurun_out/${TAG}_tests_all.log 2>&1; echo "all tests rc $?"; tail -4 gpurun_out/${TAG}_tests_all.log
run() {   # name libpath config extra-args
  local name=$1 lib=$2 cfg=$3; shift 3
  DISCO_HIP_LIB=$lib timeout 300 python bench.py --config $cfg "$@" --no-cpu-baseline > gpurun_out/${TAG}_${cfg}_$name.json 2> gpurun_out/${TAG}_${cfg}_$name.err
  python - <<P
import json
try:
    d = json.loads([l for l in open('gpurun_out/${TAG}_${cfg}_$name.json') if l.startswith('{')][-1])
    print('$cfg $* $name', round(d['ms_per_step'], 3), 'parity', d['parity_sample'] and d['parity_sample']['worst_rel'], {k: round(v['ms'], 3) for k, v in (d.get('stages') or {}).items()})
except Exception as e:
    print('$cfg $name failed', e); print(open('gpurun_out/${TAG}_${cfg}_$name.err').read()[-1200:])
P
}
NEW=$PWD/disco_amd/lib/libdisco_hip.so
run new $NEW C3 --steps 20 --warmup 3
run new $NEW C2 --steps 20 --warmup 3
run new4000 $NEW C2 --rooms 4000 --steps 10 --warmup 2
run new $NEW C5 --steps 8 --warmup 2
for l in "$@"; do run $l $PWD/exp_libs/libdisco_$l.so C5 --steps 8 --warmup 2; done
run new2 $NEW C5 --steps 8 --warmup 2
