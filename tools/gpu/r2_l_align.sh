#!/bin/bash
# stage times of C3 / C5 / C2x4000 with the hot kernels pinned to 4-KiB page boundaries (DISCO_KERNEL_ALIGN)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r02_l_align}
timeout 300 python bench.py --config C3 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_C3.json 2> gpurun_out/${TAG}_C3.err
timeout 300 python bench.py --config C5 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_C5.json 2> gpurun_out/${TAG}_C5.err
timeout 300 python bench.py --config C2 --rooms 4000 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_C2x4000.json 2> gpurun_out/${TAG}_C2x4000.err
python - <<P
import json
for c in ('C3','C5','C2x4000'):
    try:
        d = json.loads([l for l in open('gpurun_out/${TAG}_%s.json' % c) if l.startswith('{')][-1])
        print(c, round(d['ms_per_step'],3), d['parity_sample']['worst_rel'], {k: round(v['ms'],3) for k, v in d['stages'].items()})
    except Exception as e:
        print(c, 'ERR', e)
P
