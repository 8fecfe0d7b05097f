#!/bin/bash
# Round 4, pass c: float64 step-1 statistics (cov1_mode 64) against the float32 forms, room pass with 4 / 8 sub-chunks (plain float32 tree).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
timeout 600 python -m pytest tests -m gpu -q -x -k "room_cov or overlapped or iterated or cov_solve_apply" > gpurun_out/r04_c_tests_wide.log 2>&1; echo "wide tests rc $? ($(( $(date +%s) - T0 )) s)"; tail -3 gpurun_out/r04_c_tests_wide.log
T1=$(date +%s)
timeout 1500 python tools/gpu/exp_c5_variants.py gpurun_out/r04_c_c5_variants.json sample=0,25,50,75,100,125,150,175,199 variants=4:64:0:0,4:64:0:0,8:64:0:0,4:8:0:0,8:8:0:0,4:4:0:0,4:1:0:0 > gpurun_out/r04_c_c5_variants.log 2>&1; echo "variants rc $? ($(( $(date +%s) - T1 )) s)"; tail -8 gpurun_out/r04_c_c5_variants.log | cut -c1-330
echo "total $(( $(date +%s) - T0 )) s"
