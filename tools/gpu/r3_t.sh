#!/bin/bash
# Round 3, pass t: the room pass with TWO frames per barrier (six ring slots; DISCO_ROOM_FPB=2 build) against one, same box: parity first.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
DISCO_HIP_LIB=$PWD/exp_libs/libdisco_roomfpb2.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "room or iterated or wide" 2>&1 | tail -3
for v in base roomfpb2 base roomfpb2; do
if [ $v = base ]; then L=disco_amd/lib/libdisco_hip.so; else L=exp_libs/libdisco_$v.so; fi
DISCO_HIP_LIB=$PWD/$L timeout 600 python bench.py --config C5 --no-cpu-baseline > gpurun_out/r03_t_C5_$v.json 2> gpurun_out/r03_t_C5_$v.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads(open('gpurun_out/r03_t_C5_$v.json').read().strip().splitlines()[-1])
print('C5 $v', round(d['ms_per_step'], 3), 'ms', 'pipe', d['roofline']['pipeline']['frac'], 'parity', d['parity_sample']['ok'], d['parity_sample']['worst_rel'], {k: round(x['ms'], 3) for k, x in d['stages'].items()})
PY
done
