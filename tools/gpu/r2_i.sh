cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r02_i}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_zz_node_sharded_torch.py -m gpu -q -x -k "geometry or c2_single or end_to_end or reuse or node_sharded or large_batch" > gpurun_out/${TAG}_tests_sel.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/${TAG}_tests_sel.log
timeout 300 python bench.py --config C2 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_C2.json 2> gpurun_out/${TAG}_bench_C2.err; echo "bench C2 rc $?"
timeout 300 python bench.py --config C2 --rooms 4000 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_C2x4000.json 2> gpurun_out/${TAG}_bench_C2x4000.err; echo "bench C2x4000 rc $?"
python - <<PY
import json
for c in ('C2','C2x4000'):
    try:
        l=[x for x in open(f'gpurun_out/${TAG}_bench_{c}.json') if x.startswith('{')][-1]
        d=json.loads(l)
        print(c,'ms/step',round(d['ms_per_step'],3),'value',round(d['value']/1e6,2),'M nf/s xRT',round(d['x_realtime'],1), d['roofline'] and (d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['pipeline']['frac']), d['parity_sample'] and (d['parity_sample']['worst_rel'], d['parity_sample']['ok']))
        print('   ', {k:v['ms'] for k,v in (d['stages'] or {}).items()})
    except Exception as e:
        print(c,'ERR',e)
PY
