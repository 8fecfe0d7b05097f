#!/bin/bash
# two-stream overlap experiment on C3 / C5 / C2
mkdir -p gpurun_out
for c in C3 C5 C2; do
  timeout 400 python tools/gpu/exp_two_streams.py $c > gpurun_out/exp_streams_$c.log 2>&1
  tail -12 gpurun_out/exp_streams_$c.log
done
