#!/bin/bash
# Round 5, pass b: the solver's door without rounding (unscaled sums, (hi, lo) leading block in the DPP solver) against the round-4 door
# (exp_libs/libdisco_olddoor.so) on 32 of C5's 200 rooms, room_sub 8 and 4; then the solver tests on the new library.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
nproc
timeout 900 python tools/gpu/exp_c5_variants.py gpurun_out/r5_b_c5_newdoor.json sample=spread:32 variants=8:64:0:0,4:64:0:0 steps=4 2>&1 | grep -v "^$" | tail -4
DISCO_HIP_LIB=$PWD/exp_libs/libdisco_olddoor.so timeout 900 python tools/gpu/exp_c5_variants.py gpurun_out/r5_b_c5_olddoor.json sample=spread:32 variants=8:64:0:0,4:64:0:0 steps=4 2>&1 | grep -v "^$" | tail -4
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "solver or room_cov or iterated or c5_full or cov_solve" 2>&1 | tail -4
