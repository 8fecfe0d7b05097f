cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in "--rooms 1 --nodes 1" "--rooms 1 --nodes 4" "--rooms 16 --nodes 4" "--rooms 64 --nodes 4" "--rooms 250 --nodes 4"; do
timeout 200 python bench.py $cfg --steps 50 --warmup 5 --no-cpu-baseline --no-stage-timing 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg', 'ms/step %.3f' % d['ms_per_step'], 'node-frames/s %.3e' % d['value'], 'xRT %.0f' % d['x_realtime'])"
done
