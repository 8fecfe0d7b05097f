#!/bin/bash
# Round 3, pass g: swizzled k_stft_cov tile A/B against the previous build (exp_libs/libdisco_base.so) on the same box; node-sharded line.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do
for lib in new base; do
  if [ $lib = new ]; then L=""; else L="DISCO_HIP_LIB=$GRAFT_REPO_ROOT/exp_libs/libdisco_base.so"; fi
  env $L DISCO_OVERLAP_SOLVES=0 timeout 300 python bench.py --extras none --no-cpu-baseline --no-parity --steps 20 > gpurun_out/r3g_C3_$lib.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/r3g_C3_$lib.json').read().strip().splitlines()[-1]); print('C3 $lib', round(d['ms_per_step'],3), {s:x['ms'] for s,x in d['stages'].items()})"
done; done
for lib in new base; do
  if [ $lib = new ]; then L=""; else L="DISCO_HIP_LIB=$GRAFT_REPO_ROOT/exp_libs/libdisco_base.so"; fi
  env $L timeout 300 python bench.py --config C2 --rooms 4000 --extras none --no-cpu-baseline --no-parity --steps 10 > gpurun_out/r3g_C2_$lib.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/r3g_C2_$lib.json').read().strip().splitlines()[-1]); print('C2x4000 $lib', round(d['ms_per_step'],3), {s:x['ms'] for s,x in d['stages'].items()})"
done
timeout 300 python bench.py --shard nodes --rooms 250 --extras none --no-cpu-baseline > gpurun_out/r3g_nodeshard.json 2>/dev/null
python - <<'PY'
import json
for l in open('gpurun_out/r3g_nodeshard.json'):
    if l.startswith('{'):
        d = json.loads(l); print('nodeshard', round(d['ms_per_step'], 3), 'ms xRT', round(d['x_realtime'], 1), 'parity', d['parity_sample']['worst_rel_all_ranks'], 'gather ms', d['exchange']['ms_per_gather'])
PY
timeout 600 python -m pytest tests -m gpu -x -q -k "stft or end_to_end or reference or steps_state or device" 2>&1 | tail -3
