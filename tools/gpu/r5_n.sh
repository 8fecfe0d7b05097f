#!/bin/bash
# Round 5, pass n: EVERY room of bench.py's C5 batch (200 rooms x 8 x 8, 1024-pt, 2 iterations) against the float64 oracle (48 oracle processes, one thread each)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 MKL_NUM_THREADS=1; mkdir -p gpurun_out
free -g | head -2; nproc
timeout 1700 python tools/gpu/exp_c5_variants.py gpurun_out/r5_n_c5_all_200_rooms.json sample=spread:200 workers=48 variants=8:64:0:0 steps=4 2>&1 | grep -v "^$" | tail -4
