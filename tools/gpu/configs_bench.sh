# GPU tests + the other BASELINE configs (not the headline): C2, C5-shaped (staged P = 15), C4 (CRNN masks).
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/pytest_gpu.log
run() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-stage-timing "$@" > gpurun_out/cfg_$name.log 2>&1; echo "$name rc $?"; grep '^{' gpurun_out/cfg_$name.log | tail -1 | python -c "
import json,sys
l=sys.stdin.read()
if l.strip():
    d=json.loads(l); print('$name', 'ms/step %.2f' % d['ms_per_step'], 'node-frames/s %.3e' % d['value'], 'xRT %.1f' % d['x_realtime'], d['config']['workload'][:90])
else:
    print('$name no output'); print(open('gpurun_out/cfg_$name.log').read()[-1200:])
"; }
run C2 --rooms 256 --nodes 1 --mics 4 --steps 5 --warmup 2
run C2big --rooms 4000 --nodes 1 --mics 4 --steps 3 --warmup 1
run C5 --rooms 200 --nodes 8 --mics 8 --n-fft 1024 --steps 3 --warmup 1
run C4 --rooms 100 --mask crnn --steps 2 --warmup 1
