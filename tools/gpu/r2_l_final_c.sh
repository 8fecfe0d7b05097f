#!/bin/bash
# round-2 pass l, final binaries (page-aligned kernels): full GPU suite, kernel stats + PMC traffic + bench lines of C3, C5, C2
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r02_m}
timeout 900 python -m pytest tests -m gpu -q -rs > gpurun_out/${TAG}_tests_all.log 2>&1; echo "all tests rc $?"; grep -E "passed|failed|FAILED|ERROR" gpurun_out/${TAG}_tests_all.log | head -5
bash tools/profile_round.sh ${TAG}_C3 --config C3
cp gpurun_out/${TAG}_C3_pmc_traffic.json profiles/pmc_traffic_C3.json
timeout 400 python bench.py --config C3 --steps 10 --warmup 3 > gpurun_out/${TAG}_bench_C3.json 2> gpurun_out/${TAG}_bench_C3.err; echo "bench C3 rc $?"
bash tools/profile_round.sh ${TAG}_C5 --config C5
cp gpurun_out/${TAG}_C5_pmc_traffic.json profiles/pmc_traffic_C5.json
timeout 400 python bench.py --config C5 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_C5.json 2> gpurun_out/${TAG}_bench_C5.err; echo "bench C5 rc $?"
bash tools/profile_round.sh ${TAG}_C2x4000 --config C2 --rooms 4000
cp gpurun_out/${TAG}_C2x4000_pmc_traffic.json profiles/pmc_traffic_C2.json
timeout 300 python bench.py --config C2 --rooms 4000 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_C2x4000.json 2> gpurun_out/${TAG}_bench_C2x4000.err
timeout 300 python bench.py --config C2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_C2.json 2> gpurun_out/${TAG}_bench_C2.err
cp profiles/pmc_traffic_C*.json gpurun_out/
python - <<PY
import json
for c in ('C3','C5','C2x4000','C2'):
    try:
        l=[x for x in open(f'gpurun_out/${TAG}_bench_{c}.json') if x.startswith('{')][-1]
        d=json.loads(l)
        rf=d.get('roofline')
        print(c,'ms/step',round(d['ms_per_step'],3),'value',round(d['value']/1e6,2),'M nf/s xRT',round(d['x_realtime'],1), rf and (rf['kernel'], rf['frac'], rf['traffic'], rf.get('traffic_note'), rf['pipeline']['frac']), d.get('parity_sample') and (d['parity_sample']['worst_rel'], d['parity_sample']['ok']))
        print('   ', {k:v['ms'] for k,v in (d['stages'] or {}).items()})
    except Exception as e:
        print(c,'ERR',e)
PY
