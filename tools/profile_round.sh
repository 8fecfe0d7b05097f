#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel stats + the two PMC passes behind bench.py's roofline object.
# Usage: tools/profile_round.sh <tag>     -> gpurun_out/<tag>_kernel_stats.csv, gpurun_out/<tag>_pmc_raw.json
TAG=${1:-r01}
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o $TAG -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_under_rocprof.log 2>&1; echo "stats rc $?"
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc -o fetch -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-timing > gpurun_out/pmc_fetch.log 2>&1; echo "fetch rc $?"
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc -o write -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-timing > gpurun_out/pmc_write.log 2>&1; echo "write rc $?"
python tools/pmc_extract.py gpurun_out/${TAG}_pmc_raw.json FETCH_SIZE=gpurun_out/pmc/fetch_results.db WRITE_SIZE=gpurun_out/pmc/write_results.db
python tools/rocprof_summary.py gpurun_out/prof/${TAG}_results.db gpurun_out/${TAG}_kernel_stats.csv
rm -f gpurun_out/pmc/*.db gpurun_out/prof/${TAG}_results.db
grep disco gpurun_out/${TAG}_kernel_stats.csv | cut -c1-110
