cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench.log 2>&1; echo "bench rc $?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/bench.log') if x.startswith('{')][-1]
d=json.loads(l)
print('ms/step',d['ms_per_step'],'value',d['value'],'xRT',d['x_realtime'], d['roofline']['pipeline'])
PY
