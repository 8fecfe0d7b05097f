#!/bin/bash
# Round 4, pass v: k_apply_istft_wide with parts of its loop body removed (-DDISCO_WIDE_EXP bits: 1 no transform / overlap-add, 2 no filter
# arithmetic, 4 no loads; libraries built like tools/gpu/mk_room_exp.sh does for the room pass): C5 step and apply2_istft time
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for e in 0 1 2 4 3 5 6 7; do
  if [ $e = 0 ]; then L=disco_amd/lib/libdisco_hip.so; else L=exp_libs/libdisco_wideexp$e.so; fi
  DISCO_HIP_LIB=$PWD/$L timeout 200 python bench.py --config C5 --extras none --steps 4 --warmup 2 --no-cpu-baseline --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('exp $e', round(d['ms_per_step'],2), 'apply2_istft', d['stages']['apply2_istft']['ms'])"
done
