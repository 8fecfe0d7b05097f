#!/bin/bash
# Round 4, pass n: neighbouring tiles of the persistent room pass on one XCD: C5 stage times + HBM counters
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x -k "room_cov or iterated or c5_full" > gpurun_out/r04_n_tests.log 2>&1; echo "tests rc $?"; tail -2 gpurun_out/r04_n_tests.log
timeout 900 python tools/gpu/exp_c5_variants.py gpurun_out/r04_n_c5_variants.json sample=0,100,199 variants=8:64:0:0,8:64:0:0,4:64:0:0 > gpurun_out/r04_n_c5_variants.log 2>&1; echo "variants rc $?"; head -4 gpurun_out/r04_n_c5_variants.log | cut -c1-330; tail -3 gpurun_out/r04_n_c5_variants.log | cut -c1-200
bash tools/profile_round.sh r04_n_C5 --config C5 2>&1 | tail -12
