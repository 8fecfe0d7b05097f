cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in base d6 d6w2 d6r2; do
  export DISCO_HIP_LIB=$PWD/exp_libs/libdisco_$v.so
  timeout 300 python bench.py --rooms 200 --online-every 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/x_online_$v.json 2>/dev/null
  timeout 300 python bench.py --config C5 --steps 4 --warmup 1 --no-cpu-baseline --no-parity > gpurun_out/x_c5_$v.json 2>/dev/null
  timeout 300 python bench.py --config C3 --steps 6 --warmup 2 --no-cpu-baseline --no-parity > gpurun_out/x_c3_$v.json 2>/dev/null
  python - <<PY
import json
for c in ('online','c5','c3'):
    try:
        d=json.loads([x for x in open(f'gpurun_out/x_{c}_$v.json') if x.startswith('{')][-1])
        print('$v',c,'ms/step',round(d['ms_per_step'],3),'xRT',round(d['x_realtime'],1),{k:v['ms'] for k,v in (d['stages'] or {}).items() if 'solve' in k or 'online' in k or 'cov' in k})
    except Exception as e: print('$v',c,'ERR',e)
PY
done
