#!/bin/bash
# Round 4, pass p: the wide shapes' final filter + iSTFT in one pass (k_apply_istft_wide): parity, then C5 with and without it and with
# several run lengths
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "apply_istft_wide or room_cov or iterated or overlapped" > gpurun_out/r04_p_tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r04_p_tests.log
timeout 900 python tools/gpu/exp_c5_variants.py gpurun_out/r04_p_c5_variants.json variants=8:64:0:-1,8:64:0:0,8:64:0:21,8:64:0:11,8:64:0:8,8:64:0:6 sample=0,199 > gpurun_out/r04_p_c5_variants.txt 2>&1; tail -8 gpurun_out/r04_p_c5_variants.txt
