#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel stats + the two PMC passes behind bench.py's roofline object.
# Usage: tools/profile_round.sh <tag> [bench args, e.g. --config C5]
#   -> gpurun_out/<tag>_kernel_stats.csv, gpurun_out/<tag>_pmc_raw.json, gpurun_out/<tag>_pmc_traffic.json
TAG=${1:-r02}
shift
ARGS="$@"
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
DISCO_OVERLAP_SOLVES=0 timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o $TAG -- python bench.py $ARGS --extras none --steps 5 --warmup 2 --no-cpu-baseline --no-parity > gpurun_out/${TAG}_bench_under_rocprof.log 2>&1; echo "stats rc $?"
# (DISCO_OVERLAP_SOLVES=0: kernels one at a time, the mode bench.py takes its per-kernel roofline figure in)
# counters in their own runs, --kernel-trace only (the PMC + trace-domain combination is refused on this pool)
DISCO_OVERLAP_SOLVES=0 timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_$TAG -o fetch -- python bench.py $ARGS --extras none --steps 1 --warmup 0 --no-cpu-baseline --no-stage-timing --no-parity --pmc-calibrate > gpurun_out/${TAG}_pmc_fetch.log 2>&1; echo "fetch rc $?"
DISCO_OVERLAP_SOLVES=0 timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_$TAG -o write -- python bench.py $ARGS --extras none --steps 1 --warmup 0 --no-cpu-baseline --no-stage-timing --no-parity --pmc-calibrate > gpurun_out/${TAG}_pmc_write.log 2>&1; echo "write rc $?"
python tools/pmc_extract.py gpurun_out/${TAG}_pmc_raw.json FETCH_SIZE=gpurun_out/pmc_$TAG/fetch_results.db WRITE_SIZE=gpurun_out/pmc_$TAG/write_results.db
python tools/rocprof_summary.py gpurun_out/prof_$TAG/${TAG}_results.db gpurun_out/${TAG}_kernel_stats.csv
python tools/pmc_traffic.py gpurun_out/${TAG}_pmc_raw.json gpurun_out/${TAG}_pmc_traffic.json
rm -rf gpurun_out/pmc_$TAG gpurun_out/prof_$TAG
grep disco gpurun_out/${TAG}_kernel_stats.csv | cut -c1-120 | head -12
