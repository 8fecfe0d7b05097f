#!/bin/bash
# Round 4, pass h: the streaming online path on the MI355X + the whole GPU suite.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
timeout 600 python -m pytest tests -m gpu -q -x -k "online" > gpurun_out/r04_h_tests_online.log 2>&1; echo "online tests rc $? ($(( $(date +%s) - T0 )) s)"; tail -4 gpurun_out/r04_h_tests_online.log
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04_h_tests.log 2>&1; echo "tests rc $? ($(( $(date +%s) - T0 )) s)"; grep -E "passed|failed" gpurun_out/r04_h_tests.log | tail -2; grep -E "^FAILED|^E  " gpurun_out/r04_h_tests.log | head -10
