#!/bin/bash
# Round 4, pass aa: look-ahead (DISCO_COV64_AHEAD variant libraries in exp_libs/) x covariance chunks of the cooperative float64 step-1 pass
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for e in 3 2 4 6; do
  if [ $e = 3 ]; then L=disco_amd/lib/libdisco_hip.so; else L=exp_libs/libdisco_cov64a$e.so; fi
  for ch in 0 2 4; do
  DISCO_HIP_LIB=$PWD/$L timeout 200 python bench.py --config C5 --extras none --steps 4 --warmup 2 --no-cpu-baseline --no-parity --tuning 0,$ch,0,0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ahead $e chunks $ch', round(d['ms_per_step'],2), 'cov1', d['stages']['cov1']['ms'], 'solve1', d['stages']['solve1']['ms'])"
  done
done
