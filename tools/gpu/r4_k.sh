#!/bin/bash
# Round 4, pass k: the one-exchange wave transform (two 512-point transforms per wave, fft_wave_2x512) against fft_wave<512>
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 tools/gpu/kbench/fft_rate > gpurun_out/r04_k_fft_rate.txt 2>&1; echo "rc $?"; cat gpurun_out/r04_k_fft_rate.txt
