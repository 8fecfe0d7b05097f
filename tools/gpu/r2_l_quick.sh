#!/bin/bash
# round-2 pass l, quick: the allocation / hipGraph tests, then eager vs graph-replay step times on small batches
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r02_l}
timeout 600 python -m pytest tests/test_gpu_graph_capture.py tests/test_gpu_parity.py -x -q -k "graph or no_allocation or native_library" > gpurun_out/${TAG}_tests_quick.log 2>&1; echo "tests rc $?"
tail -5 gpurun_out/${TAG}_tests_quick.log
for r in 1 8 64; do
  for g in "" "--graph"; do
    timeout 200 python bench.py --config C3 --rooms $r --steps 300 --warmup 30 --no-cpu-baseline --no-stage-timing $g > gpurun_out/${TAG}_small_${r}${g}.json 2> gpurun_out/${TAG}_small_${r}${g}.err || tail -5 gpurun_out/${TAG}_small_${r}${g}.err
  done
done
timeout 200 python bench.py --config C2 --steps 200 --warmup 20 --no-cpu-baseline --no-stage-timing > gpurun_out/${TAG}_small_C2.json 2> gpurun_out/${TAG}_small_C2.err
timeout 200 python bench.py --config C2 --steps 200 --warmup 20 --no-cpu-baseline --no-stage-timing --graph > gpurun_out/${TAG}_small_C2--graph.json 2> gpurun_out/${TAG}_small_C2--graph.err
timeout 300 python bench.py --config C3 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_quick_C3.json 2> gpurun_out/${TAG}_quick_C3.err
timeout 300 python bench.py --config C3 --steps 10 --warmup 3 --no-cpu-baseline --graph --no-stage-timing > gpurun_out/${TAG}_quick_C3--graph.json 2> gpurun_out/${TAG}_quick_C3--graph.err
python - <<P
import json, glob
for f in sorted(glob.glob('gpurun_out/${TAG}_small_*.json') + glob.glob('gpurun_out/${TAG}_quick_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], 'ms/step', round(d['ms_per_step'], 4), 'M nf/s', round(d['value'] / 1e6, 2), d['config'].get('launch'), d.get('parity_sample') and d['parity_sample']['worst_rel'])
    except Exception as e:
        print(f, 'ERR', e)
P
