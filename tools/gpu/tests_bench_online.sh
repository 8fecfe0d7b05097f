cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -6 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench.log 2>&1; echo "bench rc $?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/bench.log') if x.startswith('{')][-1]
d=json.loads(l)
print('ms/step',d['ms_per_step'],'value',d['value'],'xRT',d['x_realtime'], d['roofline']['pipeline'])
print({k:v['ms'] for k,v in d['stages'].items()})
PY
for U in 1 8; do
timeout 300 python bench.py --rooms 200 --steps 2 --warmup 1 --no-cpu-baseline --online-every $U > gpurun_out/bench_online_$U.log 2>&1; echo "online $U rc $?"
python - <<PY
import json
l=[x for x in open('gpurun_out/bench_online_$U.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('online U=$U ms/step',d['ms_per_step'],'value',d['value'],'xRT',d['x_realtime'])
else:
    print(open('gpurun_out/bench_online_$U.log').read()[-1500:])
PY
done
