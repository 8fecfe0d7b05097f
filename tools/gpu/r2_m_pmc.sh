#!/bin/bash
# PMC traffic + kernel stats of C3 / C5 / C2x4000 on the final sources (comment-only edits re-stamp the digest bench.py checks)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r02_n}
bash tools/profile_round.sh ${TAG}_C3 --config C3
cp gpurun_out/${TAG}_C3_pmc_traffic.json gpurun_out/pmc_traffic_C3.json
bash tools/profile_round.sh ${TAG}_C5 --config C5
cp gpurun_out/${TAG}_C5_pmc_traffic.json gpurun_out/pmc_traffic_C5.json
bash tools/profile_round.sh ${TAG}_C2x4000 --config C2 --rooms 4000
cp gpurun_out/${TAG}_C2x4000_pmc_traffic.json gpurun_out/pmc_traffic_C2.json
