#!/bin/bash
# Round 4, pass w: k_apply_istft_wide with separate filter and transform waves: parity, C5 with run lengths
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "apply_istft_wide or iterated or overlapped or c5_full or graph" > gpurun_out/r04_w_tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r04_w_tests.log
timeout 900 python tools/gpu/exp_c5_variants.py gpurun_out/r04_w_c5_variants.json variants=8:64:0:-1,8:64:0:0,8:64:0:40,8:64:0:21,8:64:0:14,8:64:0:11 sample=0,199 > gpurun_out/r04_w_c5_variants.txt 2>&1; grep "^room_sub" gpurun_out/r04_w_c5_variants.txt
