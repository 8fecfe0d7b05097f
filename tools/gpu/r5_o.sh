#!/bin/bash
# Round 5, pass o: parity beyond the bench line's samples -- EVERY room of the C3 headline batch (1000), 500 of C2's 4000, all 125 of C4's, 96 of the online
# mode's 1000 -- each against the float64 oracle (bench.py --parity-rooms; 8 oracle processes)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
run() { tag=$1; shift; timeout 1500 python bench.py --extras none --no-cpu-baseline --no-stage-timing --steps 3 --warmup 1 "$@" > gpurun_out/r5_o_$tag.line 2> gpurun_out/r5_o_$tag.err; echo "$tag rc $?"; python tools/gpu/parity_hist.py gpurun_out/r5_o_$tag.line gpurun_out/r5_o_parity_$tag.json; }
run C3_all_1000 --parity-rooms 1000
run C2x4000_500 --config C2 --rooms 4000 --parity-rooms 500
run C4_all_125 --config C4 --parity-rooms 125
run online1_96 --rooms 1000 --online-every 1 --parity-rooms 96
rm -f gpurun_out/r5_o_*.line
