#!/bin/bash
# Round 4, pass ae: k_stft_pairs with the spectra parked in the waves' own FFT scratch + twiddles in LDS (three workgroups per CU): STFT
# parity (more than 4 channels), wide-shape tests, two C5 lines
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "stft or room_cov or iterated or c5_full or apply_istft_wide or online_stream" > gpurun_out/r04_ae_tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r04_ae_tests.log
for i in 1 2; do timeout 200 python bench.py --config C5 --extras none --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), {s: x['ms'] for s, x in d['stages'].items()}, d['parity_sample']['worst_rel_all_ranks'], d['roofline']['pipeline']['frac'])"; done | tee gpurun_out/r04_ae_c5.txt
