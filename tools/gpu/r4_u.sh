#!/bin/bash
# Round 4, pass u: the room pass with parts of its loop body removed (libraries from tools/gpu/mk_room_exp.sh 1 2 4 8 3 7): C5 step and room_cov2 time
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for e in 0 1 2 4 8 3 7; do
  if [ $e = 0 ]; then L=disco_amd/lib/libdisco_hip.so; else L=exp_libs/libdisco_roomexp$e.so; fi
  DISCO_HIP_LIB=$PWD/$L timeout 200 python bench.py --config C5 --extras none --steps 4 --warmup 2 --no-cpu-baseline --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('exp $e', round(d['ms_per_step'],2), 'room_cov2', d['stages']['room_cov2']['ms'])"
done
