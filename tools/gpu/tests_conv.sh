cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/pytest_gpu.log
timeout 300 python tools/conv_time.py 1000 4096 2>&1 | tail -1
timeout 300 python tools/conv_time.py 250 8192 2>&1 | tail -1
