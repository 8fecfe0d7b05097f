#!/bin/bash
# Round 4, pass i: SQ / LDS counters of the C5 kernels (persistent room pass, float64 step-1 statistics)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
BARGS="--config C5" bash tools/gpu/pmc_alu.sh r04_i_C5 2>&1 | grep -E "rc|disco::" | cut -c1-420
