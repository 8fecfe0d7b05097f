#!/bin/bash
# Round 4, pass b: C5 with (hi, lo) partial blocks, hand-overs per item (room_flush) and sub-chunk counts; wide-shape GPU tests first.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
timeout 600 python -m pytest tests -m gpu -q -x -k "room_cov or overlapped or iterated" > gpurun_out/r04_b_tests_wide.log 2>&1; echo "wide tests rc $? ($(( $(date +%s) - T0 )) s)"; tail -3 gpurun_out/r04_b_tests_wide.log
T1=$(date +%s)
timeout 1500 python tools/gpu/exp_c5_variants.py gpurun_out/r04_b_c5_variants.json sample=0,25,50,75,100,125,150,175,199 variants=8:4:0:2,8:4:0:2,8:4:0:1,8:4:0:4,4:4:0:2,4:4:0:1,8:8:0:2,8:1:0:2,8:4:1:2,8:8:1:2 > gpurun_out/r04_b_c5_variants.log 2>&1; echo "variants rc $? ($(( $(date +%s) - T1 )) s)"; tail -11 gpurun_out/r04_b_c5_variants.log
echo "total $(( $(date +%s) - T0 )) s"
