#!/bin/bash
# Round 4, pass e: the new GPU tests (BASELINE-shaped reference fixtures, 8-rank one-GPU bench runs, CRNN fixture on the GPU).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -q -s -k "baseline_shapes or eight_ranks or predict_masks_on_gpu or c5_full_length" > gpurun_out/r04_e_tests_new.log 2>&1; echo "tests rc $? ($(( $(date +%s) - T0 )) s)"; grep -E "passed|failed" gpurun_out/r04_e_tests_new.log | tail -2; grep -E "^FAILED|^E  |ref_signal|max \|mask|^\{|^c[23] " gpurun_out/r04_e_tests_new.log | cut -c1-300 | head -40
