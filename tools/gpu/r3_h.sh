#!/bin/bash
# Round 3, pass h: the two-rank bookkeeping test of the bench line on one GPU, smoke(), the remaining new GPU tests.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_zz_node_sharded_torch.py -x -q 2>&1 | tail -15
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
