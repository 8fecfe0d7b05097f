#!/bin/bash
# Round 5, pass c: (hi, lo) staging also in the thread solver (P = 7, 8 of the wide shapes): 32 of C5's rooms, then solver / wide-shape tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python tools/gpu/exp_c5_variants.py gpurun_out/r5_c_c5_hilo_thread.json sample=spread:32 variants=8:64:0:0,4:64:0:0 steps=4 2>&1 | grep -v "^$" | tail -4
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "solver or room_cov or iterated or c5_full or cov_solve or apply_istft_wide" 2>&1 | tail -4
