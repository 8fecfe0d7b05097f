#!/bin/bash
# round-2 pass l, part B: C2 at 4000 rooms (kernel stats + PMC traffic), bench lines of C2, C4, online, node-sharded, hipGraph replay
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r02_l}
bash tools/profile_round.sh ${TAG}_C2x4000 --config C2 --rooms 4000
cp gpurun_out/${TAG}_C2x4000_pmc_traffic.json profiles/pmc_traffic_C2.json
cp profiles/pmc_traffic_C2.json gpurun_out/
timeout 300 python bench.py --config C2 --rooms 4000 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_C2x4000.json 2> gpurun_out/${TAG}_bench_C2x4000.err
timeout 300 python bench.py --config C2 --steps 10 --warmup 3 > gpurun_out/${TAG}_bench_C2.json 2> gpurun_out/${TAG}_bench_C2.err
timeout 300 python bench.py --config C4 --steps 10 --warmup 3 > gpurun_out/${TAG}_bench_C4.json 2> gpurun_out/${TAG}_bench_C4.err
timeout 300 python bench.py --rooms 200 --online-every 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_online1.json 2> gpurun_out/${TAG}_bench_online1.err
timeout 300 python bench.py --rooms 200 --online-every 8 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_online8.json 2> gpurun_out/${TAG}_bench_online8.err
timeout 300 python bench.py --gpus 1 --shard nodes --rooms 250 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_nodeshard.json 2> gpurun_out/${TAG}_bench_nodeshard.err
timeout 300 python bench.py --config C3 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_C3_box2.json 2> gpurun_out/${TAG}_bench_C3_box2.err
timeout 300 python bench.py --config C5 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_C5_box2.json 2> gpurun_out/${TAG}_bench_C5_box2.err
timeout 300 python bench.py --config C3 --steps 10 --warmup 3 --no-cpu-baseline --no-stage-timing --graph > gpurun_out/${TAG}_bench_C3_graph.json 2> gpurun_out/${TAG}_bench_C3_graph.err
python - <<PY
import json
for c in ('C2','C2x4000','C4','online1','online8','nodeshard','C3_box2','C5_box2','C3_graph'):
    try:
        l=[x for x in open(f'gpurun_out/${TAG}_bench_{c}.json') if x.startswith('{')][-1]
        d=json.loads(l)
        rf=d.get('roofline')
        print(c,'ms/step',round(d['ms_per_step'],3),'value',round(d['value']/1e6,2),'M nf/s xRT',round(d['x_realtime'],1), rf and (rf['kernel'], rf['frac'], rf['traffic'], rf['pipeline']['frac']), d.get('parity_sample') and (d['parity_sample']['worst_rel'], d['parity_sample']['ok']))
        print('   ', {k:v['ms'] for k,v in (d.get('stages') or {}).items()})
    except Exception as e:
        print(c,'ERR',e)
PY
