cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/pytest_gpu.log
timeout 300 python tools/end_to_end_demo.py 16 /tmp/zds 2>&1 | tail -1
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms/step %.3f' % d['ms_per_step'], {k: v['ms'] for k, v in d['stages'].items()})"
