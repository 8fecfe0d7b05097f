cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/pytest_gpu.log
timeout 300 python tools/metrics_time.py 4000 160000 2>&1 | tail -2
