# usage: bash tools/gpu/profile_cfg.sh <tag> <bench args...>   -> gpurun_out/<tag>_kernel_stats.csv
TAG=$1; shift
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o $TAG -- python bench.py --no-cpu-baseline --no-stage-timing "$@" > gpurun_out/${TAG}_bench.log 2>&1; echo "rc $?"
python tools/rocprof_summary.py gpurun_out/prof/${TAG}_results.db gpurun_out/${TAG}_kernel_stats.csv
rm -f gpurun_out/prof/${TAG}_results.db
grep disco gpurun_out/${TAG}_kernel_stats.csv | cut -c1-130
grep '^{' gpurun_out/${TAG}_bench.log | tail -1 | cut -c1-300
