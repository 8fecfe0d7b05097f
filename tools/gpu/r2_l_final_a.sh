#!/bin/bash
# round-2 pass l, part A: full GPU suite, then kernel stats + PMC traffic + bench line of C3 (headline) and C5
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r02_l}
timeout 1500 python -m pytest tests -m gpu -q -rs > gpurun_out/${TAG}_tests_all.log 2>&1; echo "all tests rc $?"; grep -E "passed|failed|FAILED|ERROR" gpurun_out/${TAG}_tests_all.log | head -8
for c in C3 C5; do
  bash tools/profile_round.sh ${TAG}_$c --config $c
  cp gpurun_out/${TAG}_${c}_pmc_traffic.json profiles/pmc_traffic_$c.json
  timeout 600 python bench.py --config $c --steps 10 --warmup 3 > gpurun_out/${TAG}_bench_$c.json 2> gpurun_out/${TAG}_bench_$c.err; echo "bench $c rc $?"
done
cp profiles/pmc_traffic_C3.json profiles/pmc_traffic_C5.json gpurun_out/
python - <<PY
import json
for c in ('C3','C5'):
    try:
        l=[x for x in open(f'gpurun_out/${TAG}_bench_{c}.json') if x.startswith('{')][-1]
        d=json.loads(l)
        rf=d.get('roofline')
        print(c,'ms/step',round(d['ms_per_step'],3),'value',round(d['value']/1e6,2),'M nf/s xRT',round(d['x_realtime'],1), rf and (rf['kernel'], rf['frac'], rf['traffic'], rf.get('traffic_note'), rf['pipeline']['frac']), d.get('parity_sample') and (d['parity_sample']['worst_rel'], d['parity_sample']['ok']))
        print('   ', {k:v['ms'] for k,v in (d['stages'] or {}).items()})
    except Exception as e:
        print(c,'ERR',e)
PY
