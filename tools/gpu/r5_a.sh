#!/bin/bash
# Round 5, pass a: touch prefetch (DISCO_PF_DIST) and the Z-layout tile (DISCO_ZTILE) in k_stft_cov / k_mask_oracle / k_stft_apply_istft:
# C3 and C2 x 4000 stage times per variant library (tools/gpu/mk_variant.sh), then the parity tests of the affected kernels on the candidate.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
one() {  # lib tag, bench args
  DISCO_HIP_LIB=$1 timeout 300 python bench.py $3 --extras none --steps 6 --warmup 2 --no-cpu-baseline --no-parity 2>/tmp/err.log | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$2', '$3', 'ms/step', round(d['ms_per_step'],3), ' '.join(f\"{k}={v['ms']:.3f}\" for k,v in d['stages'].items()))" || tail -5 /tmp/err.log
}
D=$PWD/disco_amd/lib/libdisco_hip.so
one $D default "--config C3"
for v in default pf3 zt pf3zt pf2zt pf4zt default; do
  if [ $v = default ]; then L=$D; else L=$PWD/exp_libs/libdisco_$v.so; fi
  one $L $v "--config C3"
  one $L $v "--config C2 --rooms 4000"
done 2>&1 | tee gpurun_out/r5_a_variants.txt
CAND=${CAND:-pf3zt}
DISCO_HIP_LIB=$PWD/exp_libs/libdisco_$CAND.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stft or mask or end_to_end or baseline or geometry or c2_single or full_size or from_samples" 2>&1 | tail -5 | tee gpurun_out/r5_a_tests.txt
