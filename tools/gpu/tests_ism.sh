cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/pytest_gpu.log
timeout 300 python tools/ism_time.py 1000 20 2>&1 | tail -1
