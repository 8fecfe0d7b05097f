cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/gpu/c4_profile.py 125 > gpurun_out/r02_h_c4_profile.log 2>&1; echo "c4 rc $?"; grep -v "amdgpu.ids\|Warn\|warn" gpurun_out/r02_h_c4_profile.log | head -20 | cut -c1-90,150-200
timeout 600 python -m pytest tests/test_gpu_crnn_inloop.py -m gpu -q > gpurun_out/r02_h_crnn_tests.log 2>&1; echo "crnn tests rc $?"; tail -3 gpurun_out/r02_h_crnn_tests.log
timeout 600 python bench.py --config C4 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r02_h_bench_C4.json 2> gpurun_out/r02_h_bench_C4.err; echo "bench c4 rc $?"; cut -c1-400 gpurun_out/r02_h_bench_C4.json
