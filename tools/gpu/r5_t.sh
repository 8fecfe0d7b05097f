#!/bin/bash
# Round 5, pass t: the length of the float32 sums of the fused STFT + covariance pass (frames per STFT wave; a workgroup's fold sums 4 x that): speed of
# the C3 / C2x4000 step and the error of the whole-batch sweep's worst C3 rooms (profiles/r05_o_parity_C3_all_1000.json: 439, 838, 803, 682, 226)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
{
for rw in 0 40 20 10; do
  t=""; [ $rw != 0 ] && t="--tuning $rw,0,0,0"
  timeout 300 python bench.py --extras none --no-cpu-baseline --steps 10 --warmup 3 --parity-rooms 2 $t 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C3      frames per wave $rw: ms/step', round(d['ms_per_step'],3), ' '.join(f\"{k}={v['ms']:.3f}\" for k,v in d['stages'].items()))"
  timeout 300 python bench.py --config C2 --rooms 4000 --extras none --no-cpu-baseline --steps 5 --warmup 2 --parity-rooms 2 $t 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C2x4000 frames per wave $rw: ms/step', round(d['ms_per_step'],3), ' '.join(f\"{k}={v['ms']:.3f}\" for k,v in d['stages'].items()))"
done
} 2>&1 | tee gpurun_out/r5_t_runw_speed.txt
for fr in 436 836 800 680 224; do
  timeout 300 python tools/gpu/exp_c3_room.py gpurun_out/r5_t_c3_rooms_$fr.json first_room=$fr rooms=8 watch=$((fr+3)),$((fr+2)) 2>&1 | grep "frames per wave [0-9]* \|frames per wave 80 (the\|heuristic" | grep -v "chunks"
done 2>&1 | tee gpurun_out/r5_t_runw_parity.txt
