#!/bin/bash
# Round 5, pass q: C4's random-weight masks (output layer spread x 40) CLIPPED to [c, 1 - c]: all 125 rooms against the float64 oracle (same masks on both sides)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for c in 0.005 0.02 0.05; do
  DISCO_BENCH_CRNN_CLIP=$c timeout 600 python bench.py --extras none --no-cpu-baseline --no-stage-timing --steps 2 --warmup 1 --config C4 --parity-rooms 125 > gpurun_out/r5_q_$c.line 2> gpurun_out/r5_q_$c.err; echo "clip $c rc $?"
  python tools/gpu/parity_hist.py gpurun_out/r5_q_$c.line gpurun_out/r5_q_parity_C4_clip$c.json
done
rm -f gpurun_out/r5_q_*.line
