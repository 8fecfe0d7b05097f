#!/bin/bash
# Round 4, pass f: one THREAD per pencil for 5 <= P <= 8 (AGPRs as the second register file, one wave per SIMD) against the LDS group solver.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/gpu/solve_thread_time.py 1028000 5 6 7 8 > gpurun_out/r04_f_solve_thread.txt 2>&1; echo "rc $?"; cat gpurun_out/r04_f_solve_thread.txt | tail -6
