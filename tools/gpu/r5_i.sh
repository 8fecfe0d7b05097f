#!/bin/bash
# Round 5, pass i: the room pass hands its totals over as (hi, lo) blocks formed in float64 (k_room.h finish; k_solve_dpp.h stages every entry as (hi, lo)):
# 32 of C5's rooms against the float64 oracle, the wide-shape / solver / selftest GPU tests, then the default bench line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python tools/gpu/exp_c5_variants.py gpurun_out/r5_i_c5_32rooms_room_hilo.json sample=spread:32 variants=8:64:0:0 steps=4 2>&1 | grep -v "^$" | tail -4
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "solver or room or iterated or c5_full or cov_solve or apply_istft_wide or selftest or reserve or graph" 2>&1 | tail -4
timeout 600 python bench.py > gpurun_out/r5_i_bench_default.json 2> gpurun_out/r5_i_bench_default.err; tail -c 1500 gpurun_out/r5_i_bench_default.json
