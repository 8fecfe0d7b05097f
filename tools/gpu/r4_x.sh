#!/bin/bash
# Round 4, pass x: the exchange buffer of the wave FFT XOR-swizzled (conflict-free reads AND writes) instead of padded: transform rate
# (tools/gpu/kbench/fft_rate built with and without -DDISCO_FFT_XOR=0), FFT-heavy parity tests, the bench line with C2 / C2x4000 / C5
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
echo "== padded (rounds 1-3)"; timeout 200 tools/gpu/kbench/fft_rate_pad 2>&1 | tee gpurun_out/r04_x_fft_rate_pad.txt | head -30
echo "== XOR swizzle"; timeout 200 tools/gpu/kbench/fft_rate 2>&1 | tee gpurun_out/r04_x_fft_rate_xor.txt | head -30
timeout 600 python -m pytest tests -m gpu -q -x -k "stft or istft or mask or rir or end_to_end or baseline or golden" > gpurun_out/r04_x_tests.log 2>&1; echo "tests rc $?"; tail -2 gpurun_out/r04_x_tests.log
timeout 900 python bench.py --steps 10 --extras C2,C2x4000,C5 --no-cpu-baseline > gpurun_out/r04_x_bench.json 2> gpurun_out/r04_x_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04_x_bench.json').read().strip().splitlines()[-1])
print('C3', round(d['ms_per_step'], 3), {s: x['ms'] for s, x in d['stages'].items()}, 'parity', d['parity_sample']['worst_rel_all_ranks'], d['roofline'].get('pipeline', {}).get('frac'))
for k, v in d['configs'].items():
    print('  ', k, round(v['ms_per_step'], 2), {s: x['ms'] for s, x in v['stages'].items()}, 'parity', v['parity_sample']['worst_rel_all_ranks'], v['roofline'].get('pipeline', {}).get('frac'))
PY
