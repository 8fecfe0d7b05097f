# usage: bash tools/gpu/profile_and_bench.sh <tag>
TAG=${1:-r01_x}
bash tools/profile_round.sh $TAG
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/pmc_traffic.py gpurun_out/${TAG}_pmc_raw.json profiles/pmc_traffic.json > /dev/null 2>&1; cp profiles/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json
timeout 600 python bench.py > gpurun_out/${TAG}_bench_full.log 2>&1; echo "bench rc $?"; grep '^{' gpurun_out/${TAG}_bench_full.log | tail -1 | cut -c1-1500
timeout 300 python tools/metrics_time.py 4000 160000 2>&1 | tail -1
