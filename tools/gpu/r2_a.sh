# round 2, first GPU pass: solver/geometry tests first (fast feedback), then the full GPU suite, then C3/C5/C2 bench lines.
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "solver or geometry or large_batch or native" > gpurun_out/r2a_tests_new.log 2>&1; echo "new tests rc $?"; tail -5 gpurun_out/r2a_tests_new.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2a_bench_c3.json 2> gpurun_out/r2a_bench_c3.err; echo "bench c3 rc $?"; tail -2 gpurun_out/r2a_bench_c3.err
timeout 400 python bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2a_bench_c5.json 2> gpurun_out/r2a_bench_c5.err; echo "bench c5 rc $?"; tail -2 gpurun_out/r2a_bench_c5.err
timeout 300 python bench.py --config C2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2a_bench_c2.json 2> gpurun_out/r2a_bench_c2.err; echo "bench c2 rc $?"; tail -2 gpurun_out/r2a_bench_c2.err
python - <<'PY'
import json
for c in ('c3','c5','c2'):
    try:
        l=[x for x in open(f'gpurun_out/r2a_bench_{c}.json') if x.startswith('{')][-1]
        d=json.loads(l)
        print(c,'ms/step',round(d['ms_per_step'],3),'value',round(d['value']/1e6,2),'M nf/s xRT',round(d['x_realtime'],1), d['roofline'] and d['roofline']['pipeline'], d['parity_sample'] and (d['parity_sample']['worst_rel'], d['parity_sample']['ok']))
        print('   ', {k:v['ms'] for k,v in (d['stages'] or {}).items()})
    except Exception as e:
        print(c,'ERR',e)
PY
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2a_tests_all.log 2>&1; echo "all tests rc $?"; tail -5 gpurun_out/r2a_tests_all.log
