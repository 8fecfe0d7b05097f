#!/bin/bash
# Round 5, pass v: the final pass on the final sources (run 40 / fetch 4), then every room of the C3 batch against the oracle once more
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash tools/gpu/r5_k.sh 2>&1 | tail -30
timeout 1500 python bench.py --extras none --no-cpu-baseline --no-stage-timing --steps 3 --warmup 1 --parity-rooms 1000 > gpurun_out/r5_v_C3.line 2> gpurun_out/r5_v_C3.err; echo "C3 all rc $?"
python tools/gpu/parity_hist.py gpurun_out/r5_v_C3.line gpurun_out/r5_v_parity_C3_all_1000_run40.json; rm -f gpurun_out/r5_v_C3.line
