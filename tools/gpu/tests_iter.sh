cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/pytest_gpu.log
for cfg in "--rooms 200 --nodes 8 --mics 8 --n-fft 1024 --iters 2" "--rooms 1000 --iters 2"; do
timeout 300 python bench.py $cfg --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg', 'ms/step %.2f' % d['ms_per_step'], 'node-frames/s %.3e' % d['value'], 'xRT %.1f' % d['x_realtime'])"
done
