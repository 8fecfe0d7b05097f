cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lib in disco_amd/lib/libdisco_hip.so exp_libs/*.so; do
  [ -f "$lib" ] || continue
  for U in 1 8; do
  DISCO_HIP_LIB=$PWD/$lib timeout 300 python bench.py --rooms 200 --steps 2 --warmup 1 --no-cpu-baseline --online-every $U 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib U=$U ms/step %.1f  node-frames/s %.3e  xRT %.1f' % (d['ms_per_step'], d['value'], d['x_realtime']))"
  done
done
