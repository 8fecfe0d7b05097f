cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/bench_default.log 2>&1; echo "bench rc $?"; grep '^{' gpurun_out/bench_default.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k: d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','scaling','vs_baseline','dtype','data')})
print(d['config']); print(d['roofline']); print(d['cpu_baseline'])"
