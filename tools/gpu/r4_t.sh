#!/bin/bash
# Round 4, pass t (and later): wide-shape parity tests + two C5 bench lines (used for the room-pass look-ahead and the float64 step-1 pass)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "room_cov or iterated or overlapped or c5_full or apply_istft_wide" > gpurun_out/r04_t_tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r04_t_tests.log
for i in 1 2; do timeout 200 python bench.py --config C5 --extras none --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), {s: x['ms'] for s, x in d['stages'].items()}, d['parity_sample']['worst_rel_all_ranks'], d['roofline']['pipeline']['frac'])"; done | tee gpurun_out/r04_t_c5.txt
