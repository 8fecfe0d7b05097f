# A/B of library builds: bench.py once per lib in exp_libs/ plus the in-tree build.  Usage: bash tools/gpu/ab_bench.sh [extra bench args]
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lib in disco_amd/lib/libdisco_hip.so exp_libs/*.so; do
  [ -f "$lib" ] || continue
  DISCO_HIP_LIB=$PWD/$lib timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/ab_$(basename $lib .so).log 2>&1
  python - "$lib" <<'PY'
import json, sys, os
lib = sys.argv[1]
f = 'gpurun_out/ab_' + os.path.basename(lib)[:-3] + '.log'
l = [x for x in open(f) if x.startswith('{')]
if not l:
    print(lib, 'FAILED', open(f).read()[-800:])
else:
    d = json.loads(l[-1])
    print(os.path.basename(lib), 'ms/step %.3f' % d['ms_per_step'], {k: v['ms'] for k, v in (d.get('stages') or {}).items()})
PY
done
