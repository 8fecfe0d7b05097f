# bench A/B (ab_bench.sh) + the GPU parity suite against each variant library
bash tools/gpu/ab_bench.sh
cd $GRAFT_REPO_ROOT
for lib in exp_libs/*.so; do
  [ -f "$lib" ] || continue
  DISCO_HIP_LIB=$PWD/$lib timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_$(basename $lib .so).log 2>&1; echo "$lib pytest rc $?"; tail -3 gpurun_out/pytest_$(basename $lib .so).log
done
