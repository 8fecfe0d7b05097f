#!/bin/bash
# Round 5, pass h: k_stft_cov<512, 4, false> (C2's statistics pass) and k_stft_apply_istft<512, 4> (C2's filter pass) taken apart with
# timing-only builds (DISCO_SC_EXP / DISCO_SAI_EXP bits; tools/gpu/mk_variant.sh), 4000 C2-shaped rooms; stage times in ms per launch
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
one() {
  DISCO_HIP_LIB=$1 timeout 300 python bench.py --config C2 --rooms 4000 --extras none --steps 5 --warmup 2 --no-cpu-baseline --no-parity 2>/tmp/err.log | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$2', 'ms/step', round(d['ms_per_step'],3), ' '.join(f\"{k}={v['ms']:.3f}\" for k,v in d['stages'].items()))" || tail -5 /tmp/err.log
}
{
one $PWD/disco_amd/lib/libdisco_hip.so "the kernels                      "
for e in 1 2 3 8 9 11; do one $PWD/exp_libs/libdisco_scexp$e.so "stft_cov1_nostore EXP=$e"; done
for e in 1 2 3 4 8 11; do one $PWD/exp_libs/libdisco_saiexp$e.so "stft_apply_istft  EXP=$e"; done
one $PWD/disco_amd/lib/libdisco_hip.so "the kernels (again)              "
} 2>&1 | tee gpurun_out/r5_h_stft_parts.txt
