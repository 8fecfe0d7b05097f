#!/bin/bash
# Round 5, pass x: the node shard's final filter + iSTFT in one pass on the gathered z (disco_apply_istft_fused: k_apply_istft_wide shard-aware, narrow
# 4-mic shapes built) and caller-owned filter arrays in the sharded driver: GPU parity of the kernel on shards and of the unchanged wide routes, the
# one-rank node-sharded C3 step (250 rooms x 4 nodes), its kernels under rocprofv3.   Usage: r5_x.sh [tests]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
T0=$(date +%s)
if [ "$1" = tests ]; then
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_zz_node_sharded_torch.py -m gpu -q -k "apply_istft or node_sharded or c5_full or iterated" > gpurun_out/r5_x_tests.log 2>&1; tail -3 gpurun_out/r5_x_tests.log; grep -E "^FAILED|^E  " gpurun_out/r5_x_tests.log | head
echo "tests $(( $(date +%s) - T0 )) s"
fi
{
for rep in 1 2; do
timeout 300 python bench.py --shard nodes --rooms 250 --extras none --no-cpu-baseline 2>/tmp/err.log | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('nodeshard 250 rooms x 4 nodes, one rank: ms/step', round(d['ms_per_step'],3), 'parity', (d.get('parity_sample') or {}).get('worst_rel_all_ranks'))" || tail -5 /tmp/err.log
done
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r5x -o r5x -- python bench.py --shard nodes --rooms 250 --extras none --no-cpu-baseline --no-parity > /dev/null 2>&1
python tools/rocprof_summary.py gpurun_out/prof_r5x/r5x_results.db gpurun_out/r5_x_nodeshard_kernel_stats.csv 2>&1 | tail -2
grep disco gpurun_out/r5_x_nodeshard_kernel_stats.csv | cut -c1-110 | head -12
echo "total $(( $(date +%s) - T0 )) s"
} 2>&1 | tee gpurun_out/r5_x_nodeshard.txt
rm -rf gpurun_out/prof_r5x
