cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/pytest_gpu.log
bash tools/gpu/profile_cfg.sh r01_c5b --rooms 200 --nodes 8 --mics 8 --n-fft 1024 --steps 3 --warmup 1
