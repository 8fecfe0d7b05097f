cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r02_e}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_surface.py -m gpu -q -x > gpurun_out/${TAG}_tests_sel.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/${TAG}_tests_sel.log
for c in C3 C5 C2; do
  timeout 400 python bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_$c.json 2> gpurun_out/${TAG}_bench_$c.err; echo "bench $c rc $?"
done
timeout 300 python bench.py --config C2 --rooms 4000 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_C2x4000.json 2> gpurun_out/${TAG}_bench_C2x4000.err
timeout 300 python bench.py --rooms 200 --online-every 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_online1.json 2> gpurun_out/${TAG}_bench_online1.err; echo "bench online rc $?"
python - <<PY
import json
for c in ('C3','C5','C2','C2x4000','online1'):
    try:
        l=[x for x in open(f'gpurun_out/${TAG}_bench_{c}.json') if x.startswith('{')][-1]
        d=json.loads(l)
        print(c,'ms/step',round(d['ms_per_step'],3),'value',round(d['value']/1e6,2),'M nf/s xRT',round(d['x_realtime'],1), d['roofline'] and (d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['pipeline']['frac']), d['parity_sample'] and (d['parity_sample']['worst_rel'], d['parity_sample']['ok']))
        print('   ', {k:v['ms'] for k,v in (d['stages'] or {}).items()})
    except Exception as e:
        print(c,'ERR',e)
PY
