#!/bin/bash
# Round 4, pass j: float64 step-1 statistics two frames ahead, one partial block; thread solver's float64 chunk combine
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x -k "room_cov or cov_solve_apply or iterated or c5_full or solver" > gpurun_out/r04_j_tests.log 2>&1; echo "tests rc $?"; tail -2 gpurun_out/r04_j_tests.log
timeout 900 python tools/gpu/exp_c5_variants.py gpurun_out/r04_j_c5_variants.json sample=0,100,199 variants=8:64:0:0,8:64:0:0,8:8:0:0,4:64:0:0 > gpurun_out/r04_j_c5_variants.log 2>&1; echo "variants rc $?"; head -5 gpurun_out/r04_j_c5_variants.log | cut -c1-330; tail -4 gpurun_out/r04_j_c5_variants.log | cut -c1-200
