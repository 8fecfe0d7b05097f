#!/bin/bash
# Round 5, pass u: runs of 40 frames per STFT wave + four partial blocks in flight in the thread solver's fetch (the default) against the previous
# state (80, two) and against 20 / the old fetch at 40: C3, C2x4000 and C5 steps, alternating libraries, same box (tools/gpu/mk_variant.sh)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
one() {
  DISCO_HIP_LIB=$1 timeout 300 python bench.py $3 --extras none --steps $4 --warmup 3 --no-cpu-baseline --parity-rooms 2 2>/tmp/err.log | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$2', 'ms/step', round(d['ms_per_step'],3), ' '.join(f\"{k}={v['ms']:.3f}\" for k,v in d['stages'].items()), 'parity', '%.2e' % d['parity_sample']['worst_rel_all_ranks'])" || tail -5 /tmp/err.log
}
D=$PWD/disco_amd/lib/libdisco_hip.so
{
for rep in 1 2; do
one $D "C3 run 40, fetch 4 (default)" "" 12
one $PWD/exp_libs/libdisco_run80g2.so "C3 run 80, fetch 2 (before) " "" 12
one $PWD/exp_libs/libdisco_run20g4.so "C3 run 20, fetch 4          " "" 12
one $PWD/exp_libs/libdisco_run40g2.so "C3 run 40, fetch 2          " "" 12
done
one $D "C2x4000 run 40, fetch 4 (default)" "--config C2 --rooms 4000" 6
one $PWD/exp_libs/libdisco_run80g2.so "C2x4000 run 80, fetch 2 (before) " "--config C2 --rooms 4000" 6
one $D "C2x4000 run 40, fetch 4 (default)" "--config C2 --rooms 4000" 6
one $PWD/exp_libs/libdisco_run80g2.so "C2x4000 run 80, fetch 2 (before) " "--config C2 --rooms 4000" 6
one $D "C5 fetch 4 (default)" "--config C5" 5
one $PWD/exp_libs/libdisco_run80g2.so "C5 fetch 2 (before) " "--config C5" 5
one $D "C2 256 default" "--config C2" 30
one $PWD/exp_libs/libdisco_run80g2.so "C2 256 before " "--config C2" 30
} 2>&1 | tee gpurun_out/r5_u_run40_fetch4.txt
