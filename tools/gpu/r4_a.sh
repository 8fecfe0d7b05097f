#!/bin/bash
# Round 4, pass a: the persistent sub-chunked room pass on the MI355X -- wide-shape GPU tests, then C5 on one batch under the
# room_sub / cov1_sub variants (ms per step, stages, sampled rooms against the float64 oracle), then the full GPU suite.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
timeout 600 python -m pytest tests -m gpu -q -x -k "room_cov or overlapped or iterated" > gpurun_out/r04_a_tests_wide.log 2>&1; echo "wide tests rc $? ($(( $(date +%s) - T0 )) s)"; tail -3 gpurun_out/r04_a_tests_wide.log
T1=$(date +%s)
timeout 1500 python tools/gpu/exp_c5_variants.py gpurun_out/r04_a_c5_variants.json sample=0,50,100,150,199 variants=8:4:0,4:4:0,2:4:0,8:1:0,8:8:0,8:4:2 > gpurun_out/r04_a_c5_variants.log 2>&1; echo "variants rc $? ($(( $(date +%s) - T1 )) s)"; tail -8 gpurun_out/r04_a_c5_variants.log
T2=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r04_a_tests.log 2>&1; echo "tests rc $? ($(( $(date +%s) - T2 )) s)"; grep -E "passed|failed" gpurun_out/r04_a_tests.log | tail -2; grep -E "^FAILED|^E  " gpurun_out/r04_a_tests.log | head -10
echo "total $(( $(date +%s) - T0 )) s"
