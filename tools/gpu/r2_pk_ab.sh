#!/bin/bash
# Packed-arithmetic pass (csrc/pk.h): instruction-form self-test + the kernel parity tests on the new library, then an A/B of
# whole libraries (exp_libs/lib_<name>.so vs the in-tree build) on C3, the C2 shape at 4000 rooms and C5.
# usage: r2_pk_ab.sh TAG [lib names under exp_libs ...]
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r02_pk}; shift
timeout 700 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "pk_ or stft or istft or masks or step2 or cov_solve or end_to_end or geometry or c2_single or long_golden or size_independent or from_samples" > gpurun_out/${TAG}_tests_sel.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/${TAG}_tests_sel.log
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
for rep in 1 2; do
  DISCO_STEP2_FROM_SAMPLES=1 run new_fs$rep $NEW C3 --steps 20 --warmup 3
  DISCO_STEP2_FROM_SAMPLES=0 run new$rep $NEW C3 --steps 20 --warmup 3
  for l in "$@"; do run $l$rep $PWD/exp_libs/lib_$l.so C3 --steps 20 --warmup 3; done
done
run new $NEW C2 --rooms 4000 --steps 10 --warmup 2
for l in "$@"; do run $l $PWD/exp_libs/lib_$l.so C2 --rooms 4000 --steps 10 --warmup 2; done
run new $NEW C5 --steps 6 --warmup 2
for l in "$@"; do run $l $PWD/exp_libs/lib_$l.so C5 --steps 6 --warmup 2; done
